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
  D4EST_HIP_TUNE_STIFFNESS_WAVE = 1,     /* 1: single-wavefront two-buffer stiffness kernel where (deg_quad+1)^2 <= 64 */
  D4EST_HIP_TUNE_COUNT = 2
};
void d4est_hip_plan_set_tuning(d4est_hip_plan_t* plan, int key, int value);
int d4est_hip_plan_local_nodes(const d4est_hip_plan_t* plan);
int d4est_hip_plan_local_nodes_quad(const d4est_hip_plan_t* plan);
int d4est_hip_plan_n_elements(const d4est_hip_plan_t* plan);

/* Geometric factors in the reference's SoA layout (src/Mesh/d4est_mesh.h:123-169,
 * d4est_mesh.c:2757-2776): J_quad[local_nodes_quad];
 * rst_xyz_quad[(3*i+j)*local_nodes_quad + quad_stride[e] + n] = d r_i / d x_j.
 * on_device != 0: the two pointers are device pointers, else host pointers.
 * The plan keeps J and the pre-combined symmetric metric  W J (dr/dx)(dr/dx)^T
 * (6 entries per quadrature node, element-blocked); the inputs are not retained. */
void d4est_hip_plan_set_geometry(d4est_hip_plan_t* plan, const double* J_quad, const double* rst_xyz_quad, int on_device);

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
/* dudr_i = D_i u, i = 0..2 : d4est_laplacian_compute_dudr (d4est_laplacian.c:237-282), 3 applies of
 * d4est_operators_apply_dij (d4est_operators.c:1385-1410) per element. */
void d4est_hip_compute_dudr(d4est_hip_plan_t* plan, const double* u_dev, double* dudr0_dev, double* dudr1_dev, double* dudr2_dev);

/* Host-pointer convenience for a drop-in behind d4est's host double* API: copies u to the
 * device, applies, copies Au back (PCIe-inclusive; not the measured path). */
void d4est_hip_apply_stiffness_matrix_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host);

#ifdef __cplusplus
}
#endif
#endif /* D4EST_HIP_H */
