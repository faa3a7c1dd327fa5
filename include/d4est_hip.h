/*
 * d4est_hip.h -- C-ABI of the MI355X (gfx950) matrix-free DG operator-apply engine.
 *
 * Drop-in boundary for d4est's element hot path.  Plain C:
 * pointers, ints and sizes only.  Vectors are ELEMENT-ORDERED exactly like d4est's
 * (element e occupies [nodal_stride[e], nodal_stride[e] + (deg[e]+1)^3), x fastest;
 * reference: src/Mesh/d4est_mesh.c:2395-2470, src/dGMath/d4est_operators.c:1318-1323).
 *
 * Error convention follows the reference: every entry point is void / returns a
 * handle, and invalid input or a HIP failure prints "[D4EST_HIP_ABORT] ..." and
 * abort()s, as D4EST_ABORT does (src/Utilities/d4est_util.h:171).
 *
 * Unless a function says "host", every double* / int* argument named *_dev is a
 * DEVICE pointer (hipMalloc / d4est_hip_malloc / a torch CUDA tensor's data_ptr).
 * All launches go to the plan's stream (default: the null stream).
 */
#ifndef D4EST_HIP_H
#define D4EST_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d4est_hip_plan d4est_hip_plan_t;
typedef struct d4est_hip_transfer d4est_hip_transfer_t;   /* hp-multigrid inter-grid transfer, see below */

/* quadrature types (reference: [quadrature] name = legendre | lobatto,
 * src/Quadrature/d4est_quadrature_legendre.c:6-20, d4est_quadrature_lobatto.c:6-21) */
#define D4EST_HIP_QUAD_LEGENDRE 0
#define D4EST_HIP_QUAD_LOBATTO 1

/* ---- library info ---------------------------------------------------------- */
const char* d4est_hip_version(void);
int d4est_hip_device_count(void);

/* ---- 1-D operator tables (host) -------------------------------------------------
 * Replaces the lazily built cache of d4est_operators_t
 * (src/dGMath/d4est_operators.h:9-51, d4est_operators.c:196-304).  `out` is a HOST
 * buffer, row-major.  Returns the number of doubles written; with out == NULL only
 * returns the size.  deg_b is ignored by one-degree tables. */
enum d4est_hip_table_id {
  D4EST_HIP_TABLE_LOBATTO_NODES = 0,     /* deg_a+1            d4est_operators.c:727-733  */
  D4EST_HIP_TABLE_LOBATTO_WEIGHTS = 1,   /* deg_a+1            d4est_operators.c:735-741  */
  D4EST_HIP_TABLE_GAUSS_NODES = 2,       /* deg_a+1            d4est_operators.c:790-796  */
  D4EST_HIP_TABLE_GAUSS_WEIGHTS = 3,     /* deg_a+1            d4est_operators.c:798-805  */
  D4EST_HIP_TABLE_DIJ = 4,               /* N x N              d4est_operators.c:855-872  */
  D4EST_HIP_TABLE_MIJ = 5,               /* N x N              d4est_operators.c:712-724  */
  D4EST_HIP_TABLE_INVMIJ = 6,            /* N x N              d4est_operators.c:849-853  */
  D4EST_HIP_TABLE_LOBATTO_TO_GAUSS = 7,  /* (deg_b+1)x(deg_a+1) deg_a=lobatto, deg_b=gauss  d4est_operators.c:411-438 */
  D4EST_HIP_TABLE_P_PROLONG = 8,         /* (deg_b+1)x(deg_a+1) deg_a=degH, deg_b=degh      d4est_operators.c:995-1012 */
  D4EST_HIP_TABLE_HP_PROLONG = 9,        /* 2x(deg_b+1)x(deg_a+1)                            d4est_operators.c:944-993 */
  D4EST_HIP_TABLE_P_RESTRICT = 10,       /* (deg_a+1)x(deg_b+1)                              d4est_operators.c:1165-1185 */
  D4EST_HIP_TABLE_HP_RESTRICT = 11       /* 2x(deg_a+1)x(deg_b+1)                            d4est_operators.c:1232-1259 */
};
int d4est_hip_table(int table_id, int deg_a, int deg_b, double* out_host);

/* ---- device memory helpers for C hosts ------------------------------------------ */
void* d4est_hip_malloc(size_t bytes);
void d4est_hip_free(void* ptr_dev);
void d4est_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
void d4est_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
void d4est_hip_memset(void* dst_dev, int value, size_t bytes);
void d4est_hip_device_synchronize(void);

/* ---- plan -----------------------------------------------------------------------
 * One plan per (mesh, rank): mirrors what d4est_mesh_update produces
 * (src/Mesh/d4est_mesh.c:2790) -- per element deg, deg_quad, nodal_stride,
 * quad_stride (src/Mesh/d4est_element_data.h:13-48) -- all HOST int arrays of
 * length n_elements.  Elements are bucketed by (deg, deg_quad) internally. */
d4est_hip_plan_t* d4est_hip_plan_create(int n_elements, const int* deg, const int* deg_quad,
                                        const int* nodal_stride, const int* quad_stride, int quad_type);
void d4est_hip_plan_destroy(d4est_hip_plan_t* plan);
/* hipStream_t passed as void*; NULL = null stream */
void d4est_hip_plan_set_stream(d4est_hip_plan_t* plan, void* hip_stream);
/* Performance knobs (never change results beyond fp64 re-association).  Value -1 (default) = auto. */
enum d4est_hip_tuning_key {
  D4EST_HIP_TUNE_STIFFNESS_PREFETCH = 0, /* 1: request the metric at kernel entry (deg_quad <= 7), 0: at the point of use */
  D4EST_HIP_TUNE_STIFFNESS_WAVE = 1,     /* where (deg_quad+1)^2 <= 64: 0 multi-buffer kernel, 1 single-wavefront kernel, 2 two-wavefront kernel with metric prefetch, 3 single-wavefront kernel with pipelined operator loads (auto default for odd N), 11 even-odd single-wavefront kernel (auto default for even N, NQ); every value computes the same stiffness apply (tests/test_volume_gpu.py) */
  D4EST_HIP_TUNE_STIFFNESS_STAGGER = 2,  /* single-wave kernel: delay (units of 1024 cycles) of every other resident workgroup row */
  D4EST_HIP_TUNE_FLUX_FAST = 3,          /* 0: always the generic flux kernel; else the wave-per-face kernel where all degrees <= 7 */
  D4EST_HIP_TUNE_STIFFNESS_BIGP = 4,     /* p >= 8: 0 three-field kernel, 1 two-field multi-wave kernel (default except p = 15), 2 at p = 12 ... 15 (deg_quad = deg): the FP64 matrix-core kernel (v_mfma_f64_16x16x4; default at p = 15, where no padding is needed) */
  D4EST_HIP_TUNE_OVERLAP_TRACES = 5,     /* 1: the trace kernel runs on a side stream beside the volume kernel (default off: the event waits cost more) */
  D4EST_HIP_TUNE_STIFFNESS_EO = 6,       /* multi-buffer / multi-wave kernels: 0 plain contractions, else (default) even-odd contractions where deg+1 and deg_quad+1 are even */
  D4EST_HIP_TUNE_AFFINE = 7,             /* 0: always stream the per-node metric (the reference's general path); else (default) buckets whose elements all have a node-independent J (dr/dx)(dr/dx)^T (detected in plan_set_geometry, 4 ulp) rebuild the metric from 6 numbers per element */
  D4EST_HIP_TUNE_GHOST_ALIAS = 8,        /* set before plan_set_faces; 1: all ghost sides of a conforming plan share block 0 of the ghost trace buffer (for a ghost trace that is the same on every side: the zero trace of a Schwarz subdomain plan); compute_ghost_traces is then meaningless */
  D4EST_HIP_TUNE_GRAPH = 9,              /* 1: d4est_hip_cheby_iterate is captured into a hipGraph on its first call and replayed while its arguments (pointers, iteration count, eigenvalue window) stay the same -- for launch-bound meshes (multigrid coarse levels); needs a non-null plan stream and no exchange callback (single rank); default off */
  D4EST_HIP_TUNE_FUSE_UPDATE = 10,       /* 0: cheby_iterate runs its update as a separate kernel; else (default) on conforming meshes up to p = 15 the update rides in the epilogue of the flux kernel (same roundings, bit-identical) */
  D4EST_HIP_TUNE_FACE_DIRECT = 11,       /* 0: apply_aij / apply_lhs / the smoothers always run the two-phase face kernels (traces, then flux); default: conforming plans with one degree, deg_quad <= 7 and at least 768 elements (below that the two-phase kernels, with several wavefronts per element, are faster on the mostly empty chip) run the single-wavefront kernel that forms both sides' traces from u itself (no trace arrays; with ghost sides the trace kernel still feeds the exchange); 1: that kernel for the face terms only, the volume kernel separately, whatever the size; 2: the whole operator in that one kernel where deg_quad = deg (else as 1), whatever the size -- the default picks this form too.  Conforming plans with one degree deg = deg_quad = 8 ... 15 have the multi-wave form of that kernel (one workgroup per element: volume term, then the trace-free face terms; d4est_hip_direct_mw.hip), which is the default at EVERY size; values 0 / 1 / 2 select as above */
  D4EST_HIP_TUNE_STREAM = 12,            /* stream mode of the volume kernels with more than 8192 elements or deg >= 8, and of the p = 8 ... 15 whole-operator kernel: the data an apply reads or writes exactly once (metric, mortar factors, A u) moves with the non-temporal hint. default (-1): on when metric + u + A u of one apply exceed 320 MB (they then do not fit the 256 MB Infinity Cache, and keeping them out of the caches leaves those to u: level 5, p = 7 stiffness 191 -> 156 us); 0 / 1 force it off / on.  Same numbers either way */
  D4EST_HIP_TUNE_HP_SPLIT = 13,          /* set before plan_set_faces.  Plans with hanging faces, every deg and deg_quad <= 7 (any number of ranks): 0 every side through the tiled mortar-record kernels; 1 the conforming sides of the whole mesh AND the small sides of the hanging faces (one mortar each, the hanging factor folded into the geometric factors) through the fast conforming face kernels, only the big sides (four mortars) through the record kernels; default (-1): that split unless more than half of the elements would stay with the record kernels (level-4 brick, p = 7, every 64th ... every 3rd octant refined: apply_aij 203 -> 139 ... 661 -> 448 us).  Round 4: plans with degrees up to 15 take the split too (their conforming sides through the tiled conforming kernels, on the two family lists where the plan has them), and the record kernels run in their unit form (a side's sub-mortar records on four wavefronts); environment D4EST_HIP_HP_SPLIT_FAST_ONLY=1 / D4EST_HIP_NO_HANG_UNITS=1 restore round 3's forms */
  D4EST_HIP_TUNE_HYBRID = 14,            /* set before plan_set_faces.  Mixed-degree and locally refined plans (one rank): 0 every element through the two-phase kernels; CLEAN elements -- deg_quad = deg <= 15 and all six sides conforming against a local element of the same degree or the domain boundary -- get the whole operator from the trace-free one-kernel path of their degree (faces_direct_kernel / operator_mw_kernel over an element list), only the rest runs traces + volume + flux, on lists.  default (-1): where that was measured to pay -- the clean elements are one degree bucket and at least half of the mesh (locally refined meshes of one degree: level 4, p = 7, every 64th octant refined, apply_aij 144 -> 125 us); with several clean buckets the largest one is kept where it holds at least half of the mesh (one dominant degree; the others' elements stay two-phase) -- or, at size (at least 2048 clean elements per clean bucket), every bucket --, else every bucket would be its own latency-structured launch and the two-phase kernels win at these sizes; 1: whenever there is a clean element.  Same operator either way (tests/test_hybrid_gpu.py).  Hanging-aware form (plans under the hp split, elements with deg <= 7; environment D4EST_HIP_HYBRID_NO_HANGING=1 switches it off; mixed-aware form likewise for a conforming side against a lower-degree neighbour, D4EST_HIP_HYBRID_NO_MIXED=1): a hanging side does not make an element dirty -- big sides stay with the record kernels, small sides read the big element's sub-mortar block from the trace array and export their own; on a locally refined mesh of one degree EVERY element then takes the one-kernel path (level 4, p = 7, every 64th octant refined: 125 -> 81 us) and cheby_iterate carries its update in the operator kernel and the record flux kernel */
  D4EST_HIP_TUNE_COUNT = 15
};
void d4est_hip_plan_set_tuning(d4est_hip_plan_t* plan, int key, int value);
/* name of the stiffness kernel the last d4est_hip_apply_stiffness_matrix selected (for reports / profiles) */
const char* d4est_hip_plan_last_kernel(const d4est_hip_plan_t* plan);
/* Which face kernels apply_aij / apply_lhs / the smoothers run on this plan (after plan_set_faces, with the current tuning):
 * "direct" (one kernel, traces formed from u in place: conforming plans with one degree, deg_quad <= 7 or deg = deg_quad <= 15),
 * "direct+volume" (the same kernel also applies the element's volume term and writes A u once: deg_quad = deg <= 15, after
 * plan_set_geometry), "two-phase" (trace kernel, then flux kernel) or "hybrid: direct+volume on C clean elements, two-phase on D"
 * (mixed-degree / locally refined plans, tuning key D4EST_HIP_TUNE_HYBRID). */
const char* d4est_hip_plan_face_path(const d4est_hip_plan_t* plan);
int d4est_hip_plan_local_nodes(const d4est_hip_plan_t* plan);
int d4est_hip_plan_local_nodes_quad(const d4est_hip_plan_t* plan);
/* 1 when the plan runs in stream mode (tuning key D4EST_HIP_TUNE_STREAM: forced, or chosen from the plan's size), else 0 */
int d4est_hip_plan_stream_mode(const d4est_hip_plan_t* plan);
int d4est_hip_plan_n_elements(const d4est_hip_plan_t* plan);

/* Geometric factors in the reference's SoA layout (src/Mesh/d4est_mesh.h:123-169,
 * d4est_mesh.c:2757-2776): J_quad[local_nodes_quad];
 * rst_xyz_quad[(3*i+j)*local_nodes_quad + quad_stride[e] + n] = d r_i / d x_j.
 * on_device != 0: the two pointers are device pointers, else host pointers.
 * The plan keeps J and the pre-combined symmetric metric  W J (dr/dx)(dr/dx)^T
 * (6 entries per quadrature node, element-blocked); the inputs are not retained. */
void d4est_hip_plan_set_geometry(d4est_hip_plan_t* plan, const double* J_quad, const double* rst_xyz_quad, int on_device);

/* Volume factors with DX_compute_method = GEOM_COMPUTE_NUMERICAL (src/Mesh/d4est_mesh.c:2637-2671): the caller hands over only the
 * physical coordinates of the Lobatto nodes, xyz_lobatto = x[local_nodes] | y[local_nodes] | z[local_nodes] (d4est_factors->xyz, any
 * geometry: cubed sphere, disk, ...; 24 B/node instead of the 80 B per quadrature node of plan_set_geometry); the device forms
 * dx_d/dr_d1 = interpolate(D_d1 x_d) at the quadrature nodes, J and dr/dx (src/Geometry/d4est_geometry.c:877-976) and the
 * pre-combined metric.  Mortar factors still come through plan_set_mortar_geometry. */
void d4est_hip_plan_set_geometry_numerical(d4est_hip_plan_t* plan, const double* xyz_lobatto, int on_device);
/* Geometric factors generated ON THE DEVICE for the reference's `brick` geometry ([geometry] name = brick, X0..Z1;
 * src/Geometry/d4est_geometry_brick.c:140-206: dx_d/dr_d = (X1_d - X0_d) (dq / P4EST_ROOT_LEN) / 2, diagonal, constant per element) --
 * SURVEY.md section 8f rank 4, brick only.  Replaces d4est_hip_plan_set_geometry / _set_mortar_geometry on a brick: no J_quad /
 * rst_xyz_quad / mortar arrays are formed on the host or uploaded.  elem_dq[e] = the quadrant's side length in p4est integer
 * coordinates (d4est_element_data_t::dq), root_len = P4EST_ROOT_LEN, extents = {X0, X1, Y0, Y1, Z0, Z1}.  The mortar variant is
 * called where d4est_hip_plan_set_mortar_geometry would be (after plan_set_faces / plan_set_sipg); hanging faces use the mortar-sized
 * cell (src/Mesh/d4est_mortars.c:420-470), face_h_type FACE_H_EQ_J_DIV_SJ_QUAD. */
void d4est_hip_plan_set_geometry_brick(d4est_hip_plan_t* plan, const int* elem_dq, double root_len, const double* extents);
void d4est_hip_plan_set_mortar_geometry_brick(d4est_hip_plan_t* plan, const int* elem_dq, double root_len, const double* extents);

/* Geometric factors of an ANALYTIC tree map generated on the device (DX_compute_method =
 * GEOM_COMPUTE_ANALYTIC, JAC_compute_method = GEOM_COMPUTE_NUMERICAL as the reference requires, src/Mesh/d4est_mesh.c:2637-2680,
 * :858-1108): the host hands over only where every element sits in the forest -- d4est_element_data_t::tree, ::q[3], ::dq
 * (src/Mesh/d4est_element_data.h:13-48), p4est integer coordinates, root_len = P4EST_ROOT_LEN -- instead of 96 B per quadrature node
 * and 24 doubles per mortar node.  geom_type / params:
 *   D4EST_HIP_GEOM_CUBED_SPHERE_7TREE  [geometry] name = cubed_sphere_7tree (src/Geometry/d4est_geometry_cubed_sphere.c:498-580,
 *                                      :1884-1899): params = {R0, R1, compactify_inner_shell}; trees 0..5 wedges, 6 the centre cube
 * The mortar variant is called where d4est_hip_plan_set_mortar_geometry would be (after plan_set_hanging / plan_set_faces /
 * plan_set_sipg), needs the same three arrays for the ghost elements (order of ghost_deg), follows faces between trees through
 * side_reorder / side_orientation and hanging faces through the half-size virtual children of the big element
 * (src/Mesh/d4est_mortars.c:419-468); face_h_type FACE_H_EQ_J_DIV_SJ_QUAD. */
#define D4EST_HIP_GEOM_CUBED_SPHERE_7TREE 1
void d4est_hip_plan_set_geometry_analytic(d4est_hip_plan_t* plan, int geom_type, const double* params, const int* elem_tree,
                                          const int* elem_q, const int* elem_dq, double root_len);
void d4est_hip_plan_set_mortar_geometry_analytic(d4est_hip_plan_t* plan, int geom_type, const double* params, const int* elem_tree,
                                                 const int* elem_q, const int* elem_dq, const int* ghost_tree, const int* ghost_q,
                                                 const int* ghost_dq, double root_len);

/* ---- volume kernels (device vectors of local_nodes doubles) ---------------------- */
/* Au = K u : replaces d4est_laplacian_apply_stiffness_matrix (src/dGMath/d4est_laplacian.c:198-234)
 * = loop of d4est_quadrature_apply_stiffness_matrix (src/Quadrature/d4est_quadrature.c:263-382).
 * Au is OVERWRITTEN (as the reference's zero-fill at :337). */
void d4est_hip_apply_stiffness_matrix(d4est_hip_plan_t* plan, const double* u_dev, double* Au_dev);
/* Mu = M u : loop of d4est_quadrature_apply_mass_matrix (src/Quadrature/d4est_quadrature.c:385-477) */
void d4est_hip_apply_mass_matrix(d4est_hip_plan_t* plan, const double* u_dev, double* Mu_dev);
/* out = V^T W J f_quad : d4est_quadrature_apply_galerkin_integral (d4est_quadrature.c:142-213);
 * f_quad_dev has local_nodes_quad doubles (element e at quad_stride[e]) */
void d4est_hip_apply_galerkin_integral(d4est_hip_plan_t* plan, const double* f_quad_dev, double* out_dev);
/* u_quad = V u : d4est_quadrature_interpolate (d4est_quadrature.c:966-1016) */
void d4est_hip_interpolate(d4est_hip_plan_t* plan, const double* u_dev, double* u_quad_dev);
/* out = V^T (W J c) V u : d4est_quadrature_apply_fofufofvlilj with QUAD_APPLY_MATRIX (d4est_quadrature.c:593-774) =
 * d4est_quadrature_apply_mass_matrix with jac_quad replaced by jac_quad * f(u) f(v).  The reference evaluates the
 * callbacks f(u), f(v) on the host at the quadrature nodes (:661-683); here the caller hands their product
 * coeff_quad_dev (local_nodes_quad doubles, element e at quad_stride[e]) -- e.g. d4est_hip_interpolate + one
 * elementwise kernel of its own. */
void d4est_hip_apply_weighted_mass_matrix(d4est_hip_plan_t* plan, const double* u_dev, const double* coeff_quad_dev, double* out_dev);
/* out = V^-1 (W J)^-1 V^-T in : d4est_quadrature_apply_inverse_mass_matrix (d4est_quadrature.c:1222-1331).  As in the
 * reference this is always Gauss-Legendre and needs deg_quad == deg on every element (the assert at :1233); aborts
 * otherwise. */
void d4est_hip_apply_inverse_mass_matrix(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev);
/* out = (M (x) M (x) M) in and (M^-1 (x) M^-1 (x) M^-1) in per element, M the 1-D reference mass matrix:
 * d4est_operators_apply_mij / d4est_operators_apply_invmij (d4est_operators.c:891-928), batched over the plan. */
void d4est_hip_apply_mij(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev);
void d4est_hip_apply_invmij(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev);
/* out = D_dir in and out = D_dir^T in per element, dir = 0, 1, 2 = r, s, t: d4est_operators_apply_dij / _dij_transpose
 * (d4est_operators.c:1385-1410, :2259-2284), batched over the plan; in and out must not alias. */
void d4est_hip_apply_dij(d4est_hip_plan_t* plan, const double* in_dev, int dir, double* out_dev);
void d4est_hip_apply_dij_transpose(d4est_hip_plan_t* plan, const double* in_dev, int dir, double* out_dev);
/* Trace of a volume field on face `face` of every element and its inverse scatter: d4est_operators_apply_slicer / _apply_lift
 * (d4est_operators.c:1521-1582, :1454-1519), batched.  A face vector holds N_e^2 values per element (tangential axes in increasing
 * order, the first fastest), element e at sum_{e' < e} N_{e'}^2; d4est_hip_plan_face_nodes = its length.  The lift zero-fills. */
int d4est_hip_plan_face_nodes(const d4est_hip_plan_t* plan);
void d4est_hip_apply_slicer(d4est_hip_plan_t* plan, const double* in_dev, int face, double* out_face_dev);
void d4est_hip_apply_lift(d4est_hip_plan_t* plan, const double* in_face_dev, int face, double* out_dev);
/* dudr_i = D_i u, i = 0..2 : d4est_laplacian_compute_dudr (d4est_laplacian.c:237-282), 3 applies of
 * d4est_operators_apply_dij (d4est_operators.c:1385-1410) per element. */
void d4est_hip_compute_dudr(d4est_hip_plan_t* plan, const double* u_dev, double* dudr0_dev, double* dudr1_dev, double* dudr2_dev);

/* ---- faces: SIPG mortar terms (conforming mortars, Dirichlet boundaries) ------------------------------
 * Flat side list, side s = 6*e + f ((-) element e, face f = 0..5 = -x,+x,-y,+y,-z,+z), HOST int arrays of 6*n_elements:
 * what the reference's face iteration hands to its flux callback (src/Mesh/d4est_mortars.c:601-803):
 *   side_nbr[s]        >= 0: local (+) element; -1: domain boundary; <= -2: ghost element g = -(v+2)
 *   side_nbr_face[s]   face of the (+) element
 *   side_reorder[s]    flip0 | flip1<<1 | transpose<<2, the result of p4est_expand_face_transform as used by
 *                      d4est_operators_reorient_face_data (src/dGMath/d4est_operators.c:2031-2081); 0 inside one tree
 *   side_mortar_stride[s]  scalar offset S of the side's mortar quadrature data (d4est_laplacian_flux.c:417-449)
 *   side_bndry_stride[s]   offset of the side's Dirichlet values (boundary sides)
 * Ghost elements: ghost_deg / ghost_deg_quad (n_ghost), as p4est_ghost_exchange_data ships them (Mesh/d4est_ghost.c:52). */
void d4est_hip_plan_set_faces(d4est_hip_plan_t* plan, const int* side_nbr, const int* side_nbr_face, const int* side_reorder,
                              const int* side_mortar_stride, const int* side_bndry_stride, int total_mortar_nodes,
                              int total_bndry_nodes, int n_ghost, const int* ghost_deg, const int* ghost_deg_quad);
/* The side arrays of d4est_hip_plan_set_faces / _set_hanging built on the HOST from a p8est connectivity and the list of quadrants
 * alone -- for hosts without p4est_iterate; a d4est build records the same numbers from its face callback.
 * tree_to_tree / tree_to_face: p8est connectivity (6 per tree; face + 6 * orientation; a boundary face points at itself).  Local
 * elements: tree, q (3 per element, p4est integer coordinates in [0, root_len)), dq (side length), deg, deg_quad -- in the rank's element
 * order; ghost quadrants (the off-rank face neighbours the rank knows) likewise, referenced as -(g + 2).  The mesh must be 2:1
 * balanced across faces.  Outputs (caller-allocated: 6 n_local ints each, side_nbr4 24 n_local): everything set_faces / set_hanging
 * take, with mortar strides assigned in side order (the small sides of a hanging face share the block of their group's first local
 * member, src/Mesh/d4est_mesh.c:956-962) -- to be used with device-generated factors (plan_set_mortar_geometry_brick / _analytic) or
 * with host arrays laid out by the same strides.  Returns 1 when the mesh has hanging faces (then call plan_set_hanging). */
int d4est_hip_build_sides(int n_trees, const int* tree_to_tree, const int* tree_to_face, int root_len, int n_local, const int* tree,
                          const int* q, const int* dq, const int* deg, const int* deg_quad, int n_ghost, const int* ghost_tree,
                          const int* ghost_q, const int* ghost_dq, const int* ghost_deg_quad, int* side_nbr, int* side_nbr_face,
                          int* side_reorder, int* side_orientation, int* side_hang, int* side_sub, int* side_nbr4,
                          int* side_mortar_stride, int* side_bndry_stride, int* total_mortar_nodes, int* total_bndry_nodes);
/* The integer topology tables the library works with (csrc/d4est_hip_topology.h), for hosts that want the same numbers and for the
 * pin against the reference's own data (tests/test_topology_tables.py): id 0 p8est_face_corners [6][4], 1 p8est_face_dual [6],
 * 2 p8est_face_permutations [8][4], 3 p8est_face_permutation_sets [3][4], 4 p8est_face_permutation_refs [6][6], 5 p8est_corner_faces [8][3]
 * (p4est-2.8 src/p8est_connectivity.c:29-63, :145-152); 10 / 11 / 12 d4est_reference_p8est_FToF_code [6][6] / _code_to_perm [3][4] /
 * _perm_to_order [8][4] (src/dGMath/d4est_reference.c:3-12).  Returns the entry count (row-major ints written to out; out == NULL: count
 * only), -1 for an unknown id. */
int d4est_hip_topology_table(int id, int* out);
/* Non-conforming (hanging, 1 <-> 4) faces of a 2:1 balanced mesh; call BEFORE d4est_hip_plan_set_faces (conforming meshes skip it).
 * HOST int arrays over the sides s = 6*e + f, mirroring the two calls the reference's face iteration makes per hanging face
 * (src/Mesh/d4est_mortars.c:700-803: (e_m[4], faces_m = 4 | e_p[1]) and (e_m[1] | e_p[4], faces_p = 4)):
 *   side_hang[s]        0 conforming or boundary; 1 "big" side: this face is split, (+) side = 4 small elements;
 *                       2 "small" side: this element is one of the 4 hanging elements, (+) side = the big element (side_nbr[s])
 *   side_sub[s]         small side: index of this element among the 4 (p4est order of the hanging quadrants = z-order on the face)
 *   side_nbr4[4s..4s+3] big side: the four (+) elements in (-) order (e_p_oriented, src/Mesh/d4est_element_data.c:130-150);
 *                       small side: the four members e_m[0..3] of its own group
 *   side_orientation[s] p4est face orientation 0..3 (selects d4est_reference_reorient_face_order, dGMath/d4est_reference.c:84-110)
 * Mortar data layout as the reference allocates it (src/Mesh/d4est_mesh.c:956-979): the block of a hanging face holds its 4
 * sub-mortars one after another (scalars), vector / matrix components are strided by the block's TOTAL node count, the four small
 * sides share ONE block (the same side_mortar_stride), and drst_dxyz_p_porder is stored in the (+) side's sub-face order.
 * Element references (side_nbr, side_nbr4) are local ids or ghost codes -(g + 2), as in side_nbr.  Plans with hanging faces
 * and ghost elements take the ghost traces from the trace exchange (the *_sub block accessors below); d4est_hip_compute_ghost_traces
 * (whole ghost elements) serves conforming plans only.
 * Faces between trees: side_orientation / side_reorder as p4est reports them; every (f_m, f_p, orientation) triple is followed as the
 * reference computes it.  For transposed pairs with exactly one flip seen from the lower-numbered face (reorder codes 5, 6 -- none in
 * the reference's own connectivities) d4est_operators_reorient_face_data is not the geometric map; on a SMALL side of such a pair the
 * reference takes u and du/dx from different children of the big face, which the engine reproduces when the four sub-mortars share one
 * quadrature degree and the big element is local, and aborts otherwise. */
void d4est_hip_plan_set_hanging(d4est_hip_plan_t* plan, const int* side_hang, const int* side_sub, const int* side_nbr4,
                                const int* side_orientation);
/* SIPG parameters ([flux] sipg_penalty_prefactor, sipg_penalty_fcn; d4est_laplacian_flux_sipg.c:945-1005):
 * fcn 0 maxp_sqr_over_minh (default), 1 meanp_sqr_over_meanh, 2 maxpp1_sqr_over_minh, 3 mean_p_sqr_over_h.
 * Call BEFORE d4est_hip_plan_set_mortar_geometry (the penalty is folded into the face factors). */
void d4est_hip_plan_set_sipg(d4est_hip_plan_t* plan, double penalty_prefactor, int penalty_fcn);
/* Mortar geometric factors in the reference's layout (src/Mesh/d4est_mesh.c:946-1108): with T nodes on the side's mortar,
 * sj[S+k], n[3S + d*T + k], drst_dxyz_m / drst_dxyz_p_porder[9S + (i+3j)*T + k] = d r_i/d x_j, hm[S+k], hp[S+k]. */
void d4est_hip_plan_set_mortar_geometry(d4est_hip_plan_t* plan, const double* sj, const double* n, const double* drst_dxyz_m,
                                        const double* drst_dxyz_p_porder, const double* hm, const double* hp, int on_device);
/* Dirichlet values on the Lobatto face nodes of every boundary side (EVAL_BNDRY_FCN_ON_LOBATTO,
 * d4est_laplacian_flux_sipg.c:80-112); NULL resets to zero (the homogeneous operator used by apply_lhs). */
void d4est_hip_plan_set_dirichlet_values(d4est_hip_plan_t* plan, const double* g_lobatto, int on_device);
/* Robin boundary condition on ALL boundary sides instead of Dirichlet (BC_ROBIN of d4est_laplacian_flux_new;
 * d4est_laplacian_flux_sipg_robin, d4est_laplacian_flux_sipg.c:339-489): the side adds  V^T W sj (coeff u_m - rhs), lifted.
 * The reference evaluates the callbacks robin_coeff / robin_rhs at the boundary mortar quadrature nodes (:388-412); here the
 * caller hands the two arrays, indexed like sj (side_mortar_stride[s] + k, total_mortar_nodes doubles; only the boundary
 * sides' entries are read).  coeff_quad == NULL switches back to Dirichlet.  Call after plan_set_mortar_geometry. */
void d4est_hip_plan_set_robin_values(d4est_hip_plan_t* plan, const double* coeff_quad, const double* rhs_quad, int on_device);
/* Trace buffers.  For every side s the engine keeps u and du/dr_{0,1,2} INTERPOLATED TO THE SIDE'S MORTAR QUADRATURE
 * NODES (what d4est_laplacian_flux_interface forms at src/dGMath/d4est_laplacian_flux.c:635-815, once per side instead
 * of once per flux call): a block of 4 T doubles, T = (deg_mortar_quad+1)^2, field c at c*T, node a + NQ*b in the
 * side's own face ordering.  Local blocks are in side order; the ghost buffer holds one block per local side whose (+)
 * element is a ghost (the (+) element's trace on ITS face), also in side order. */
long long d4est_hip_plan_trace_size(const d4est_hip_plan_t* plan);
long long d4est_hip_plan_ghost_trace_size(const d4est_hip_plan_t* plan);
/* ghost blocks from whole-element ghost data packed in ghost order (what d4est_ghost_data_exchange delivers,
 * src/Mesh/d4est_ghost_data.c:143-256); a trace exchange (RCCL) fills the same buffer directly. */
void d4est_hip_compute_ghost_traces(d4est_hip_plan_t* plan, const double* u_ghost_dev, double* ghost_trace_dev);
/* local traces of u into trace_dev (d4est_hip_plan_trace_size doubles) */
void d4est_hip_compute_face_traces(d4est_hip_plan_t* plan, const double* u_dev, double* trace_dev);
/* Au += mortar terms, given local (and ghost) traces: d4est_laplacian_apply_mortar_matrices (d4est_laplacian.c:285-315) */
void d4est_hip_apply_flux(d4est_hip_plan_t* plan, const double* trace_dev, const double* ghost_trace_dev, double* Au_dev);
/* Au = A u : d4est_laplacian_apply_aij (src/dGMath/d4est_laplacian.c:318-417) = stiffness + traces + flux.
 * ghost_trace_dev may be NULL when the plan has no ghost elements. */
void d4est_hip_apply_aij(d4est_hip_plan_t* plan, const double* u_dev, const double* ghost_trace_dev, double* Au_dev);

/* rhs = M f - A(0): d4est_laplacian_build_rhs_with_strong_bc (src/dGMath/d4est_laplacian.c:16-140), which every Problem calls once per
 * solve: the source term integrated against the test functions -- f given at the Lobatto nodes (f_on_quad = 0: M f,
 * d4est_quadrature_apply_mass_matrix per element, INIT_FIELD_ON_LOBATTO) or at the quadrature nodes (f_on_quad = 1: V^T W J f,
 * d4est_quadrature_apply_galerkin_integral, INIT_FIELD_ON_QUAD) -- minus the Laplacian applied to u = 0 WITH the inhomogeneous boundary
 * data currently set on the plan (d4est_hip_plan_set_dirichlet_values / _set_robin_values: the reference's
 * flux_fcn_data_for_build_rhs; reset them to the homogeneous data of apply_lhs afterwards).  Needs plan_set_faces; on plans with ghost
 * sides the exchange hooks of plan_set_comm.  f_dev / rhs_dev: device arrays; the _host form takes the reference's host vectors. */
void d4est_hip_build_rhs_with_strong_bc(d4est_hip_plan_t* plan, const double* f_dev, int f_on_quad, double* rhs_dev);
void d4est_hip_build_rhs_with_strong_bc_host(d4est_hip_plan_t* plan, const double* f_host, int f_on_quad, double* rhs_host);

/* Linearised nonlinear problems: the reference's apply_lhs is d4est_laplacian_apply_aij plus, per element,
 * d4est_quadrature_apply_fofufofvlilj(u_e; f(x, u0)) added with axpy 1.0 (e.g. constant_density_star_apply_jac,
 * src/Problems/ConstantDensityStar/constant_density_star_fcns.h:777-850 with :528-603).  coeff_quad_dev[local_nodes_quad] = f at
 * the quadrature nodes (device array; its values are CAPTURED by this call -- a plan-owned copy, and w J c pre-combined for the operator
 * kernels, whose volume stage then carries the term for one extra stream of 8 B per node -- so call it again whenever u0, i.e. f,
 * changes; the caller's array is not read afterwards); NULL = pure Laplacian.
 * The term then is part of d4est_hip_apply_lhs, _cheby_iterate, _cg_eigs and _schwarz_smooth; set on a Schwarz subdomain plan
 * (same array: the copies' quad_stride alias it) it is part of the subdomain operator.  d4est_hip_apply_aij stays the Laplacian. */
void d4est_hip_plan_set_lhs_coefficient(d4est_hip_plan_t* plan, const double* coeff_quad_dev);
/* ---- the multigrid MATRIX OPERATOR: the zeroth-order term on the coarse levels (csrc/d4est_hip_mgmatrix.hip) -------------------------
 * With use_matrix_operator = 1 (e.g. constant_density_star_mgpc_newton_petsc.c:591-602) the reference holds the term
 * V^T W J f(x, u0) V as ONE DENSE BLOCK PER ELEMENT on the finest level (d4est_solver_multigrid_matrix_setup_fofufofvlilj_operator,
 * src/Solver/d4est_solver_multigrid_matrix_operator.c:160-245: d4est_quadrature_apply_fofufofvlilj with QUAD_COMPUTE_MATRIX), restricts
 * the blocks level by level with the Galerkin product  sum_children P_c^T M_c P_c  (the restriction callback :6-48 ->
 * d4est_operators_compute_PT_mat_P, src/dGMath/d4est_operators.c:608-667), and the smoother's apply_lhs on every level below the finest
 * adds M_e u_e per element (constant_density_star_apply_jac_add_nonlinear_term_using_matrix,
 * src/Problems/ConstantDensityStar/constant_density_star_fcns.h:485-527, selected at :806-850 when matrix_op->matrix != matrix_at0).
 * Block layout = the reference's matrix_op->matrix: element e's (deg_e+1)^3 x (deg_e+1)^3 row-major block, blocks consecutive in
 * element order (d4est_mesh_get_local_matrix_nodes doubles in all).
 * The plan carries the term in ONE of three forms (each setter replaces the others; NULL switches its own form off):
 *   d4est_hip_plan_set_lhs_coefficient     the coefficient field (finest level; fused into the operator kernels)
 *   d4est_hip_plan_set_lhs_element_blocks  dense blocks (the reference's coarse-level form; 8 (deg+1)^3 bytes per DoF per apply)
 *   d4est_hip_plan_set_lhs_galerkin_chain  the same Galerkin operator applied matrix-free through the transfer objects and the FINE
 *                                          plan's coefficient: T_0^T .. T_{k-1}^T (V^T W J c V) T_{k-1} .. T_0 u -- 8 bytes per fine
 *                                          quadrature node per apply instead of the blocks (cheaper than blocks while
 *                                          fine quadrature nodes < (deg+1)^6 coarse entries, i.e. for up to two h-levels at equal p)
 * all three are part of d4est_hip_apply_lhs, _cheby_iterate, _cg_eigs; coefficient and blocks also of the Schwarz subdomain operator
 * (on a subdomain plan pass block_offset_host: the block of copy k is the block of its mesh element). */
/* d4est_mesh_get_local_matrix_nodes: sum over the elements of (deg+1)^6 */
long long d4est_hip_plan_matrix_nodes(const d4est_hip_plan_t* plan);
/* QUAD_COMPUTE_MATRIX for every element (d4est_quadrature.c:748-760, :1143-1186: the weighted mass matrix applied to the unit vectors,
 * column by column): blocks_dev[d4est_hip_plan_matrix_nodes] = V^T (W J coeff) V per element; coeff_quad_dev == NULL: the mass matrix */
void d4est_hip_compute_weighted_mass_blocks(d4est_hip_plan_t* plan, const double* coeff_quad_dev, double* blocks_dev);
/* blocks_dev stays the caller's (read at every apply); block_offset_host (n_elements, in doubles) or NULL = consecutive */
void d4est_hip_plan_set_lhs_element_blocks(d4est_hip_plan_t* plan, const double* blocks_dev, const long long* block_offset_host);
/* transfers[0]: this plan's level <-> the next finer level, ..., transfers[n-1]: <-> fine_plan's level; fine_plan has its geometry and
 * its coefficient (d4est_hip_plan_set_lhs_coefficient) set; the objects stay the caller's and must outlive the plan's use of them;
 * n_transfers = 0 switches the form off.  Runs on this plan's stream. */
void d4est_hip_plan_set_lhs_galerkin_chain(d4est_hip_plan_t* plan, int n_transfers, d4est_hip_transfer_t* const* transfers,
                                           d4est_hip_plan_t* fine_plan);
/* ---- smoother inner loops (device resident) -----------------------------------------------------------
 * Communication hooks for plans with ghost elements / several ranks (replace the reference's MPI calls:
 * d4est_ghost_data_exchange, src/Mesh/d4est_ghost_data.c:143-256, and sc_allreduce, d4est_solver_cg_eigs.c:181-243).
 * exchange(ctx, phase, trace_dev, ghost_trace_dev): phase 0 = post the face-trace exchange (the local trace buffer is
 * complete on the plan's stream), phase 1 = make the plan's stream wait until ghost_trace_dev is filled.
 * allreduce(ctx, scalars_dev, n): in-place SUM over ranks of n device doubles, ordered on the plan's stream. */
typedef void (*d4est_hip_exchange_fn)(void* ctx, int phase, const double* trace_dev, double* ghost_trace_dev);
typedef void (*d4est_hip_allreduce_fn)(void* ctx, double* scalars_dev, int n);
void d4est_hip_plan_set_comm(d4est_hip_plan_t* plan, d4est_hip_exchange_fn exchange, d4est_hip_allreduce_fn allreduce, void* ctx);
/* Au = A u using the plan-owned trace buffers and the communication hooks (apply_lhs of the Poisson problems,
 * src/Problems/Poisson/poisson_sinx_fcns.h:110-128). */
void d4est_hip_apply_lhs(d4est_hip_plan_t* plan, const double* u_dev, double* Au_dev);
/* d4est_solver_multigrid_smoother_cheby_iterate_aux (src/Solver/d4est_solver_multigrid_smoother_cheby.c:81-176):
 * iter Chebyshev iterations on u for A u = rhs with eigenvalue window [lmin, lmax]; r receives the residual
 * rhs - A u when compute_residual_at_end == 1 (else alpha (rhs - A u) of the last iteration, as in the reference).
 * u, rhs, Au (work), r are device vectors of local_nodes doubles. */
void d4est_hip_cheby_iterate(d4est_hip_plan_t* plan, double* u_dev, const double* rhs_dev, double* Au_dev, double* r_dev, int iter,
                             double lmin, double lmax, int compute_residual_at_end);
/* one fused Chebyshev update  r = alpha (rhs - Au); p = r + beta p; u += p  (smoother_cheby.c:135-153) */
void d4est_hip_cheby_update(d4est_hip_plan_t* plan, int n, const double* rhs_dev, const double* Au_dev, double alpha, double beta,
                            double* r_dev, double* p_dev, double* u_dev);
/* cg_eigs (src/Solver/d4est_solver_cg_eigs.c:116-275): imax CG iterations started from u (which they advance, as in the
 * reference), returns the Gershgorin bound of the Lanczos tridiagonal (use_new selects :36-64 over :9-33).
 * history_host (optional, 2*imax doubles) receives alpha_0..alpha_{imax-1}, beta_0..beta_{imax-1}. */
double d4est_hip_cg_eigs(d4est_hip_plan_t* plan, double* u_dev, const double* rhs_dev, double* Au_dev, int imax, int use_new,
                         double* history_host);
/* pack / unpack of face-trace blocks for the ghost exchange: dst[dst_off[b]+i] = src[src_off[b]+i], i < len[b];
 * the three index arrays are DEVICE arrays of n_blocks entries; runs on the plan's stream.  Replaces the per-mirror
 * memcpy loop of d4est_ghost_data_exchange (src/Mesh/d4est_ghost_data.c:196-236). */
void d4est_hip_copy_blocks(d4est_hip_plan_t* plan, int n_blocks, const double* src_dev, const long long* src_off_dev,
                           double* dst_dev, const long long* dst_off_dev, const int* len_dev);
/* offset / length (in doubles) of side s' block in the local trace buffer, and the offset of the block that side s
 * RECEIVES in the ghost buffer (-1 when its (+) element is not a ghost): the send / receive lists of an exchange */
long long d4est_hip_plan_trace_offset(const d4est_hip_plan_t* plan, int side);
long long d4est_hip_plan_ghost_trace_offset(const d4est_hip_plan_t* plan, int side);
int d4est_hip_plan_trace_block_len(const d4est_hip_plan_t* plan, int side);
/* Plans with hanging faces: a big side owns FOUR blocks (one per sub-mortar, in (-) order), every other side one.  Block `sub` of
 * side `side`: where it sits in the local trace buffer, how long it is, and -- if the element across that mortar is a ghost --
 * where its counterpart is expected in the ghost trace buffer (-1 otherwise).  The counterpart of a big side's block i is the single
 * block of small element i's side; the counterpart of a small side's block is the big element's block
 * d4est_reference_reorient_face_order(f_m, f_p, orientation, side_sub).  With sub = 0 these equal the three functions above on
 * conforming plans. */
int d4est_hip_plan_side_blocks(const d4est_hip_plan_t* plan, int side);
/* d4est_reference_reorient_face_order (dGMath/d4est_reference.c:84-110), face_dim = 2: index in the (+) side's own order of the
 * sub-face that is i in (-) order */
int d4est_hip_reorient_face_order(int f_m, int f_p, int orientation, int i);
/* The side_reorder code of a tree-boundary face pair: what d4est_operators_reorient_face_data derives through
 * p4est_expand_face_transform(min(f_m, f_p), 6*orientation + max(f_m, f_p)) (dGMath/d4est_operators.c:2031-2050):
 * flip0 | flip1 << 1 | (not aligned) << 2.  orientation = p4est's tree_to_face[.] / 6 = p4est_iter_face_info_t::orientation
 * (0 inside a tree).  Lets a C host fill side_reorder without p4est's transform tables. */
int d4est_hip_face_reorder_code(int f_m, int f_p, int orientation);
long long d4est_hip_plan_trace_offset_sub(const d4est_hip_plan_t* plan, int side, int sub);
long long d4est_hip_plan_ghost_trace_offset_sub(const d4est_hip_plan_t* plan, int side, int sub);
int d4est_hip_plan_trace_block_len_sub(const d4est_hip_plan_t* plan, int side, int sub);
/* deterministic device dot product; result_dev is a device double */
void d4est_hip_vec_dot(d4est_hip_plan_t* plan, int n, const double* x_dev, const double* y_dev, double* result_dev);

/* ---- RCCL transport of the ghost exchange (csrc/d4est_hip_comm.hip) --------------------------------------------------------------
 * The C replacement of d4est_ghost_data_exchange (src/Mesh/d4est_ghost_data.c:143-256) and of the sc_allreduce calls of cg_eigs
 * (src/Solver/d4est_solver_cg_eigs.c:181-243), one process per GPU, RCCL over xGMI.  librccl is opened at run time.
 * Communicator: rank 0 calls d4est_hip_comm_get_unique_id, the host broadcasts the d4est_hip_comm_unique_id_bytes() bytes by whatever
 * it has (MPI_Bcast in a d4est build, torch.distributed in the tests), every rank calls d4est_hip_comm_create (ncclCommInitRank;
 * collective; the calling thread's current HIP device is the rank's GPU).
 * One communicator, ONE stream: every RCCL operation of a communicator (the grouped send / receive of the trace exchange, the all-reduce
 * of the CG scalars, d4est_hip_comm_sendrecv / _allreduce_sum) is issued on a stream the communicator owns, with an event in from and an
 * event out to the calling plan's stream -- the order RCCL sees is the host's issue order on every rank, whatever streams the plans use. */
typedef struct d4est_hip_comm d4est_hip_comm_t;
typedef struct d4est_hip_rccl_exchange d4est_hip_rccl_exchange_t;
int d4est_hip_comm_unique_id_bytes(void);
void d4est_hip_comm_get_unique_id(void* id_out);
d4est_hip_comm_t* d4est_hip_comm_create(const void* unique_id, int rank, int world);
/* as above, but returns NULL (and prints the RCCL error) when ncclCommInitRank fails, instead of aborting */
d4est_hip_comm_t* d4est_hip_comm_try_create(const void* unique_id, int rank, int world);
void d4est_hip_comm_destroy(d4est_hip_comm_t* comm);
int d4est_hip_comm_rank(const d4est_hip_comm_t* comm);
int d4est_hip_comm_size(const d4est_hip_comm_t* comm);
/* ncclCommCount of the communicator: the rank count RCCL itself reports (reports / self-checks) */
int d4est_hip_comm_nccl_count(const d4est_hip_comm_t* comm);
/* Wire a plan (faces set) to the communicator: installs C exchange / allreduce hooks (no callback into the host language), so that
 * d4est_hip_apply_lhs, _cheby_iterate, _cg_eigs run on N ranks.  Per neighbouring rank p (peer_rank[p]) the blocks
 * [send_first[p], send_first[p+1]) of (send_off, send_len): offsets / lengths in doubles into the plan's LOCAL trace buffer
 * (d4est_hip_plan_trace_offset_sub / _trace_block_len_sub), and [recv_first[p], recv_first[p+1]) of (recv_off, recv_len) into the GHOST
 * trace buffer (d4est_hip_plan_ghost_trace_offset_sub); both ends list the shared faces in the same canonical order, the totals per
 * peer pair must agree.  Per apply: one pack kernel, one grouped ncclSend / ncclRecv round on a communication stream (beside the
 * volume kernel), one unpack kernel.  All arrays are HOST arrays, copied.  The returned object must outlive the plan's use of it. */
d4est_hip_rccl_exchange_t* d4est_hip_plan_set_rccl_exchange(d4est_hip_plan_t* plan, d4est_hip_comm_t* comm, int n_peers, const int* peer_rank,
                                                            const int* send_first, const long long* send_off, const int* send_len,
                                                            const int* recv_first, const long long* recv_off, const int* recv_len);
void d4est_hip_rccl_exchange_destroy(d4est_hip_rccl_exchange_t* x);
long long d4est_hip_rccl_exchange_count(const d4est_hip_rccl_exchange_t* x);          /* exchanges posted so far */
long long d4est_hip_rccl_exchange_send_doubles(const d4est_hip_rccl_exchange_t* x);   /* doubles sent / received per exchange */
long long d4est_hip_rccl_exchange_recv_doubles(const d4est_hip_rccl_exchange_t* x);
/* one grouped point-to-point round on already packed device buffers (peer p: send_dev[send_first[p]..send_first[p+1]), likewise
 * recv), ordered on the plan's stream -- the whole-element exchanges of the Schwarz smoother; and an in-place SUM over ranks */
void d4est_hip_comm_sendrecv(d4est_hip_comm_t* comm, d4est_hip_plan_t* plan, int n_peers, const int* peer_rank, const double* send_dev,
                             const long long* send_first, double* recv_dev, const long long* recv_first);
void d4est_hip_comm_allreduce_sum(d4est_hip_comm_t* comm, d4est_hip_plan_t* plan, double* scalars_dev, int n);

/* ---- host-pointer entries: the drop-in behind d4est's host double* API (SURVEY.md section 7 "hard part") -----------------------
 * Every reference caller hands over HOST vectors.  These entries take host pointers, move the data through plan-owned
 * persistent pinned staging and device mirrors (allocated once on first use -- no per-call hipMalloc), run the device-resident
 * routine and copy the results back: ONE upload and ONE download per call, however many operator applies happen inside
 * (a 15-iteration Chebyshev smoother call moves 2 vectors up and 2 down for 16 applies).  They return after the results are
 * in the host arrays.  PCIe-inclusive: not the measured path. */
void d4est_hip_apply_stiffness_matrix_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host);
/* d4est_laplacian_apply_aij on host vectors (the Laplacian alone) / apply_lhs (+ the zeroth-order term, exchange hooks) */
void d4est_hip_apply_aij_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host);
void d4est_hip_apply_lhs_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host);
/* d4est_hip_cheby_iterate on host vectors: u_host in/out, r_host out; Au_host (optional, may be NULL) receives the last A u */
void d4est_hip_cheby_iterate_host(d4est_hip_plan_t* plan, double* u_host, const double* rhs_host, double* Au_host, double* r_host,
                                  int iter, double lmin, double lmax, int compute_residual_at_end);
/* d4est_hip_cg_eigs on host vectors: u_host in/out (advanced by the CG iterations, as in the reference) */
double d4est_hip_cg_eigs_host(d4est_hip_plan_t* plan, double* u_host, const double* rhs_host, double* Au_host, int imax, int use_new,
                              double* history_host);
/* only J_quad (mass / galerkin / weighted-mass / inverse-mass kernels need no dr/dx) */
void d4est_hip_plan_set_jacobian(d4est_hip_plan_t* plan, const double* J_quad, int on_device);
/* pinned host memory and stream-ordered copies for C hosts that keep their own staging */
void* d4est_hip_host_alloc(size_t bytes);
void d4est_hip_host_free(void* ptr_host);
void d4est_hip_memcpy_h2d_async(d4est_hip_plan_t* plan, void* dst_dev, const void* src_host, size_t bytes);
void d4est_hip_memcpy_d2h_async(d4est_hip_plan_t* plan, void* dst_host, const void* src_dev, size_t bytes);
void d4est_hip_plan_synchronize(d4est_hip_plan_t* plan);

/* ---- hp-multigrid inter-grid transfer ---------------------------------------------------
 * The V-cycle's restriction / prolongation callbacks (src/Solver/d4est_solver_multigrid_callbacks.h:100-200, :245-330) walk the
 * coarse grid and call, per coarse element, d4est_operators_apply_p_prolong / _hp_prolong (prolongation) or their transposes
 * (restriction of residuals; src/dGMath/d4est_operators.c:1091-1132, :1689-1749).  A transfer object is that walk as a flat list:
 * item k has hrefine[k] = 0 (one fine element of degree degh[8k] <-> coarse element of degree degH[k]; equal degrees copy) or 1
 * (eight children in z-order with degrees degh[8k..8k+7] <-> their parent); d4est's third case (an element that is not coarsened,
 * copied child by child) is hrefine = 0 with degh = degH.  Both vectors are element-ordered and contiguous in item order, like the
 * reference's fine_stride / coarse_stride.  degh >= degH as the reference asserts (d4est_operators.c:379).  Degrees up to 17: the
 * restriction kernels hold three (deg+1)^3 fields in the 160 KB LDS; d4est_hip_transfer_create aborts above that. */
d4est_hip_transfer_t* d4est_hip_transfer_create(int n_items, const int* hrefine, const int* degH, const int* degh);
void d4est_hip_transfer_destroy(d4est_hip_transfer_t* t);
void d4est_hip_transfer_set_stream(d4est_hip_transfer_t* t, void* hip_stream);
long long d4est_hip_transfer_coarse_nodes(const d4est_hip_transfer_t* t);
long long d4est_hip_transfer_fine_nodes(const d4est_hip_transfer_t* t);
/* x_fine = P x_coarse (d4est_operators_apply_p_prolong / _hp_prolong per item) */
void d4est_hip_transfer_prolong(d4est_hip_transfer_t* t, const double* x_coarse_dev, double* x_fine_dev);
/* x_coarse = P^T x_fine (d4est_operators_apply_p_prolong_transpose / _hp_prolong_transpose per item; overwrites x_coarse) */
void d4est_hip_transfer_restrict(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev);
/* x_coarse = L2 projection of x_fine (d4est_operators_apply_p_restrict / _hp_restrict per item, src/dGMath/d4est_operators.c:1205-1230,
 * :1275-1297: M_H^-1 P^T M_h, children summed; the restriction of FIELDS, e.g. of the solution when the mesh is coarsened) */
void d4est_hip_transfer_project(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev);

/* The restriction of the multigrid matrix operator's element blocks through this transfer's item list (the restriction callback of
 * src/Solver/d4est_solver_multigrid_matrix_operator.c:6-48 for every coarse element): coarse block k = sum_c P_c^T M_c P_c over the item's
 * children (d4est_operators_compute_PT_mat_P, src/dGMath/d4est_operators.c:608-667); blocks consecutive in traversal order on both grids
 * (fine_matrix_stride / coarse_matrix_stride); an item that is a copy (hrefine 0, degh = degH) copies its block.
 * literal_window = 0: the Galerkin product as written above.  literal_window = 1: the reference's arithmetic to the letter -- at :651 it
 * takes child c's left factor as a window of the transposed STACKED prolongation (&PT[stride_P] read as (degH+1)^3 x (degh_c+1)^3), which
 * equals P_c^T for one child but interleaves rows of different children for eight; a d4est build therefore holds THAT block on
 * h-coarsened levels (non-symmetric).  The chain form above is the exact product by construction. */
long long d4est_hip_transfer_fine_matrix_nodes(const d4est_hip_transfer_t* t);
long long d4est_hip_transfer_coarse_matrix_nodes(const d4est_hip_transfer_t* t);
void d4est_hip_transfer_galerkin_blocks(d4est_hip_transfer_t* t, const double* fine_blocks_dev, double* coarse_blocks_dev, int literal_window);

/* ---- additive Schwarz smoother ------------------------------------------------------------
 * Replaces d4est_solver_schwarz_iterate (src/Solver/d4est_solver_schwarz.c:172-285) with its CG subdomain solver
 * (src/Solver/d4est_solver_schwarz_subdomain_solver_cg.c:101-249) and the Laplacian subdomain operator
 * (src/Solver/d4est_solver_schwarz_laplacian_ext.c:167-358), all subdomains of the rank at once.
 *
 * Metadata in the reference's terms (src/Solver/d4est_solver_schwarz_metadata.h:19-62), flattened: subdomain i owns the entries
 * [sub_first[i], sub_first[i+1]) of sub_elem (local element id of each subdomain element, sorted by (tree, quadid) like
 * d4est_solver_schwarz_metadata.c:447-455), sub_faces[3k..3k+2] (element_metadata.faces: the faces of the subdomain element that
 * touch the core, -1 = none; all -1 on the core) and sub_core_faces[3k..3k+2] (element_metadata.core_faces, the mirrored faces).
 * num_nodes_overlap is the [d4est_solver_schwarz] input of that name (1 .. min deg + 1).
 *
 * subdomain_plan is a plan whose elements are the subdomain elements in that order (deg / deg_quad of the mesh element; quad_stride
 * and the mortar strides of the mesh element, so the geometric factors are the mesh's own arrays; neighbours = the copies inside the
 * same subdomain; a face whose neighbour is outside the subdomain = a ghost side, which the smoother feeds with a zero trace: the
 * reference's zero_and_skip rule, src/dGMath/d4est_laplacian_flux.c:486-520, :944-962; domain boundary = -1 with homogeneous
 * Dirichlet data).  disco4est_amd/schwarz.py builds it; the plan stays owned by the caller and must outlive the handle.
 * Hanging 1 <-> 4 faces: the copies carry the mesh's plan_set_hanging arrays, group / neighbour entries remapped the same way.
 * Several ranks: the mesh handed over is the rank's EXTENDED mesh (own elements followed by the ghost layer, P4EST_CONNECT_FULL), subdomains
 * exist for the own elements only; the residual of the ghost-layer elements arrives by a whole-element exchange before schwarz_iterate and
 * their part of u (the corrections) is sent back to the owners afterwards (disco4est_amd/schwarz.py: SchwarzShard, parallel.ElementSchedule). */
typedef struct d4est_hip_schwarz d4est_hip_schwarz_t;
d4est_hip_schwarz_t* d4est_hip_schwarz_create(d4est_hip_plan_t* subdomain_plan, int n_subdomains, const int* sub_first,
                                              const int* sub_elem, const int* sub_faces, const int* sub_core_faces,
                                              int num_nodes_overlap, int n_mesh_elements, const int* mesh_deg,
                                              const int* mesh_nodal_stride);
void d4est_hip_schwarz_destroy(d4est_hip_schwarz_t* sz);
/* schwarz_metadata->nodal_size / ->restricted_nodal_size (d4est_solver_schwarz_metadata.c:459-520) */
long long d4est_hip_schwarz_nodal_size(const d4est_hip_schwarz_t* sz);
long long d4est_hip_schwarz_restricted_nodal_size(const d4est_hip_schwarz_t* sz);
/* How many element copies have their rows of the subdomain operator kept as small dense blocks (read off the matrix-free operator by
 * probing, once, on first use) instead of being applied element by element: on a conforming one-degree mesh with a small overlap
 * the corner copies (8 of 27 per subdomain; 8 restricted nodes of 512 at overlap 2, p = 7).  0 when the optimisation does not apply
 * (mixed degrees, hanging faces, blocks above 8 KB) or is switched off (D4EST_HIP_SCHWARZ_CONDENSE=0); results agree to rounding. */
int d4est_hip_schwarz_condensed_copies(d4est_hip_schwarz_t* sz);
/* d4est_solver_schwarz_convert_nodal_field_to_restricted_field_over_subdomains (src/Solver/d4est_solver_schwarz_helpers.c:123-155).
 * Fields over the subdomains have nodal_size entries (whole elements); the restricted field is stored in place, zero outside the
 * overlap nodes (restrict-transpose, helpers.c:210-239, is then the identity). */
void d4est_hip_schwarz_restrict_field(d4est_hip_schwarz_t* sz, const double* field_dev, double* out_over_subdomains_dev);
/* d4est_solver_schwarz_laplacian_ext_apply_over_subdomain for every subdomain: out = R A R^T in (in must be zero outside the overlap) */
void d4est_hip_schwarz_apply_over_subdomains(d4est_hip_schwarz_t* sz, const double* in_dev, double* out_dev);
/* d4est_solver_schwarz_compute_correction + ..._add_corrections (helpers.c:421-451, transfer_ghost_data.c:97-120):
 * u += sum over subdomains of weights * du, added in ascending subdomain order */
void d4est_hip_schwarz_add_correction(d4est_hip_schwarz_t* sz, const double* du_over_subdomains_dev, double* u_dev);
/* d4est_solver_schwarz_iterate: u += correction of the residual r = rhs - A u (both mesh vectors on the device); the three
 * [d4est_solver_schwarz] CG options as arguments.  Returns the number of batched CG sweeps (= the largest iteration count). */
int d4est_hip_schwarz_iterate(d4est_hip_schwarz_t* sz, double* u_dev, const double* r_dev, int subdomain_iter, double subdomain_atol,
                              double subdomain_rtol);
/* The multigrid smoother built on it, d4est_solver_multigrid_smoother_schwarz (src/Solver/d4est_solver_multigrid_smoother_schwarz.c:98-196):
 * smoother_iterations times { r = rhs - A u; schwarz_iterate(u, r) }, then r = rhs - A u.  mesh_plan is the plan of the mesh itself (faces
 * set, homogeneous Dirichlet data, same stream as the subdomain plan); on several ranks its apply_lhs hooks do the trace exchange, the
 * whole-element exchanges around schwarz_iterate are the host's (SchwarzShard), so this entry is for one rank. */
void d4est_hip_schwarz_smooth(d4est_hip_schwarz_t* sz, d4est_hip_plan_t* mesh_plan, double* u_dev, const double* rhs_dev, double* r_dev,
                              int smoother_iterations, int subdomain_iter, double subdomain_atol, double subdomain_rtol);
/* schwarz->subdomain_solve_iterations / _residuals of the last iterate (d4est_solver_schwarz.c:259-260), host arrays of n_subdomains */
void d4est_hip_schwarz_get_info(d4est_hip_schwarz_t* sz, int* final_iter_host, double* final_res_host);

#ifdef __cplusplus
}
#endif
#endif /* D4EST_HIP_H */
