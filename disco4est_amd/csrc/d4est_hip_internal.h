// Internal declarations shared by the HIP translation units of libd4est_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/d4est_hip.h"

#define D4EST_HIP_ABORT(...)                                   \
  do {                                                         \
    std::fprintf(stderr, "[D4EST_HIP_ABORT] ");                \
    std::fprintf(stderr, __VA_ARGS__);                         \
    std::fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__);    \
    std::abort();                                              \
  } while (0)

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) D4EST_HIP_ABORT("%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

namespace d4est_hip {

// One (deg, deg_quad) bucket of elements; launches are per bucket so N and NQ are
// compile-time constants inside the kernels.
struct Bucket {
  int deg = 0, deg_quad = 0;
  int N = 0, NQ = 0;
  int n_elem = 0;
  int elem_offset = 0;      // offset into Plan::d_elem_ids / d_ns_list / d_qs_list
  // device 1-D tables (row-major)
  double* d_B = nullptr;    // NQ x N  interpolation Lobatto -> quadrature nodes
  double* d_G = nullptr;    // NQ x N  G = B * D  (derivative evaluated at quadrature nodes)
  double* d_D = nullptr;    // N x N   collocation derivative
  double* d_w = nullptr;    // NQ      quadrature weights
  double* d_BT = nullptr;   // N x NQ  transposes (row i = column i of the operator: one scalar load feeds NQ FMA chains)
  double* d_GT = nullptr;
  double* d_DT = nullptr;
  // square N x N tables for the nodal mass applies (d4est_operators.c:891-928) and the Gauss inverse mass
  // (d4est_quadrature.c:1222-1331; only when deg_quad == deg, else null)
  // affine strides of the bucket-ordered element list (ns = ns0 + i*ns_stride); ns_stride < 0: not affine, use the lists
  int ns0 = 0, ns_stride = -1, qs0 = 0, qs_stride = -1;
  bool affine = false;  // every element has a node-independent J (dr/dx)(dr/dx)^T (detected in plan_set_geometry)
  // even-odd tables (see stiffness_wave_eo_kernel), only when N and NQ are even: forward B, G (N/2 rows of NQ) and
  // backward B^T, G^T (NQ/2 rows of N)
  double* d_EBf = nullptr;
  double* d_EGf = nullptr;
  double* d_EBb = nullptr;
  double* d_EGb = nullptr;
  // N = NQ only: even-odd tables of the differentiation matrix ON the quadrature nodes and of its transpose (collocated-gradient form)
  double* d_EDq = nullptr;
  double* d_EDqT = nullptr;
  double* d_M = nullptr;
  double* d_MT = nullptr;
  double* d_Minv = nullptr;
  double* d_MinvT = nullptr;
  double* d_Binv = nullptr;   // (lobatto_to_gauss_interp)^-1
  double* d_BinvT = nullptr;  // its transpose = (lobatto_to_gauss_interp_trans)^-1
  double* d_wGL = nullptr;    // Gauss weights
};

}  // namespace d4est_hip

struct d4est_hip_plan {
  int n_elements = 0;
  int local_nodes = 0;
  int local_nodes_quad = 0;
  int quad_type = 0;
  hipStream_t stream = nullptr;
  int n_cus = 0;  // multiProcessorCount of the device the plan was created on
  char last_kernel[128] = "";  // name of the stiffness kernel selected by the last apply (largest bucket last)

  std::vector<int> deg, deg_quad, nodal_stride, quad_stride;  // host copies
  std::vector<d4est_hip::Bucket> buckets;

  std::vector<int> elem_ids;      // element ids sorted by bucket (host)
  int* d_elem_ids = nullptr;      // same on the device
  int* d_ns_list = nullptr;       // nodal_stride in bucket order (one load, wave-uniform when 1 element/wave)
  int* d_qs_list = nullptr;       // quad_stride in bucket order

  bool has_geometry = false;
  double* d_J = nullptr;          // local_nodes_quad  (reference layout)
  int* d_face_stride = nullptr;       // per element: offset of its N^2 face values in a face vector (apply_slicer / apply_lift)
  int face_nodes = 0;
  double* d_metric = nullptr;
  double* d_metric_affine = nullptr;  // 6 per element (bucket order): J (dr/dx)(dr/dx)^T of node 0
  int* d_nonaffine = nullptr;         // per bucket: set by the precombine kernel when an element is not affine     // 6 * local_nodes_quad, element-blocked: [e][c][n], c in (rr,rs,rt,ss,st,tt)

  // ---- faces (d4est_hip_faces.hip) ----
  bool has_faces = false, has_face_geometry = false;
  int n_ghost = 0;
  std::vector<int> ghost_deg, ghost_deg_quad;
  std::vector<int> side_nbr, side_nbr_face, side_reorder, side_mortar_stride, side_bndry_stride;  // host copies, 6*n_elements
  std::vector<int> side_hang, side_sub, side_nbr4, side_orientation;  // hanging faces (d4est_hip_plan_set_hanging); empty = conforming
  int total_mortar_nodes = 0, total_bndry_nodes = 0;
  long long local_trace_doubles = 0, ghost_trace_doubles = 0;
  int* d_side_desc = nullptr;        // SideDesc per side (see d4est_hip_faces.hip)
  void* d_elem_desc = nullptr;       // ElemDesc per element
  // per mortar record (hanging plans; empty on conforming plans where a side has exactly one block)
  std::vector<long long> rec_qoff, rec_goff;
  std::vector<int> rec_len, side_first_rec;
  std::vector<long long> trace_offset, ghost_trace_offset;  // per SIDE: offset of its 4 T mortar-node trace block (ghost: -1 if none)
  double* d_face_ops = nullptr;      // concatenated 1-D face operators (C and E matrices)
  double* d_face_geom = nullptr;     // 7 * total_mortar_nodes: am[3], ap[3], s3 per mortar quadrature node (side-blocked)
  double* d_bndry = nullptr;         // Dirichlet values on boundary Lobatto face nodes (total_bndry_nodes), zero by default
  double* d_trace = nullptr;         // plan-owned local trace buffer
  double sipg_prefactor = 10.0;
  int sipg_penalty_fcn = 0;
  int max_face_lds_doubles = 0;
  bool face_fast = false;  // all sides have N, Np, NQ <= 8: flux_wave_kernel applies
  void* direct = nullptr;  // DirectHost (d4est_hip_direct.hip): conforming uniform-degree plans, deg_quad <= 7
  void* hybrid = nullptr;  // HybridHost (d4est_hip_direct.hip): mixed-degree / locally refined plans -- clean elements through the direct kernels

  // ---- solver workspace / communication hooks (d4est_hip_solver.hip) ----
  double *d_work_p = nullptr, *d_work_d = nullptr, *d_work_r = nullptr, *d_reduce = nullptr, *d_ghost_trace = nullptr;
  const double* d_lhs_coeff = nullptr;   // optional zeroth-order term of apply_lhs: + V^T W J c V u (the caller's array; non-null = term on)
  double* d_lhs_c = nullptr;             // its values as they were at plan_set_lhs_coefficient (plan-owned copy)
  double* d_lhs_wjc = nullptr;           // w J c at the quadrature nodes (one stream for the operator kernels' volume stage)
  bool lhs_wjc_valid = false;
  double* d_work_m = nullptr;            // scratch of that term
  // the same term on a coarse multigrid level (d4est_hip_mgmatrix.hip): dense element blocks (the caller's array; non-null = this form
  // is on) with one offset per element, or the Galerkin chain through transfer objects up to a fine plan's coefficient
  const double* d_lhs_blocks = nullptr;
  long long* d_lhs_block_off = nullptr;
  void* lhs_chain = nullptr;             // d4est_hip::LhsChain
  hipStream_t side_stream = nullptr;  // the trace kernel (and the exchange) run here, concurrently with the volume kernel
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // D4EST_HIP_TUNE_GRAPH: the last cheby_iterate call captured as a hipGraph (replayed while the arguments stay the same)
  hipGraphExec_t cheby_graph = nullptr;
  struct { const void *u, *rhs, *Au, *r; int iter, flag; double lmin, lmax; hipStream_t stream; } cheby_graph_key = {};
  d4est_hip_exchange_fn exchange_fn = nullptr;
  d4est_hip_allreduce_fn allreduce_fn = nullptr;
  void* comm_ctx = nullptr;

  // bumped by every call that can change what the operator computes (geometry, faces, SIPG parameters, boundary data, the zeroth-order
  // coefficient, tuning): objects that cache something derived from the operator (the Schwarz smoother's condensed blocks) compare it
  unsigned long long op_generation = 0;
  bool quad_aliased = false;      // some elements share a quadrature block (quad_stride repeats)
  int stream_mode = 0;            // 1: non-temporal metric / factor loads and A u stores (capi: update_stream_mode; kernels: with_ld)
  bool bc_inhomogeneous = false;   // non-zero Dirichlet data or Robin data is set (an affine, not linear, operator)

  int tuning[D4EST_HIP_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // -1 = auto  // see d4est_hip_plan_set_tuning

  // generic-path scratch (allocated lazily)
  double* d_scratch = nullptr;
  size_t scratch_doubles = 0;

  // host-pointer entry points (d4est_hip_*_host): persistent pinned staging (3 vectors) and device mirrors u, rhs, Au, r,
  // allocated once on first use -- no per-call hipMalloc behind d4est's host double* API
  double* h_stage = nullptr;
  double* d_host[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace d4est_hip {
struct TreeMapParams;
struct CellDesc;
void launch_analytic_geometry(d4est_hip_plan* plan, const TreeMapParams& P, const CellDesc* d_cells, double root_len);
void faces_set_geometry_analytic(d4est_hip_plan* plan, const TreeMapParams& P, const std::vector<CellDesc>& elem,
                                 const std::vector<CellDesc>& ghost, double root_len);

// d4est_hip_volume.hip
void launch_metric_precombine(d4est_hip_plan* plan, const double* d_J, const double* d_rst);
void launch_stiffness(d4est_hip_plan* plan, const double* u, double* Au);
void launch_stiffness_view(d4est_hip_plan* plan, const double* u, double* Au, int* ns_list_view, int* qs_list_view, const int* view_offset,
                           const int* view_count);
// mode: 0 mass, 1 galerkin, 2 interpolate / square tensor apply, 3 weighted mass, 4 inverse mass;
// which: 0 quadrature interpolation, 1 inverse Gauss interpolation, 2 M (1-D mass), 3 M^-1
void launch_mass_like(d4est_hip_plan* plan, int mode, const double* in, double* out, const double* coeff = nullptr, int which = 0);
void launch_brick_geometry(d4est_hip_plan* plan, const int* d_elem_dq, double root_len, const double* extents);
void launch_numerical_geometry(d4est_hip_plan* plan, const double* d_xyz);
void faces_set_geometry_brick(d4est_hip_plan* plan, const int* d_elem_dq, double root_len, const double* extents);
void launch_slicer_lift(d4est_hip_plan* plan, const double* in, double* out, int face, int lift);
void launch_dij(d4est_hip_plan* plan, const double* in, double* out, int dir, int transpose);
void launch_dudr(d4est_hip_plan* plan, const double* u, double* d0, double* d1, double* d2);

// d4est_hip_faces.hip
void faces_setup(d4est_hip_plan* plan);
void faces_set_geometry(d4est_hip_plan* plan, const double* sj, const double* n, const double* drst_m, const double* drst_p,
                        const double* hm, const double* hp, int on_device);
int reorient_face_order(int f_m, int f_p, int o, int i);  // dGMath/d4est_reference.c:84-110
int face_reorder_code(int f_m, int f_p, int o);            // dGMath/d4est_operators.c:2031-2050
void faces_set_dirichlet(d4est_hip_plan* plan, const double* g_lobatto, int on_device);
void faces_set_robin(d4est_hip_plan* plan, const double* coeff_quad, const double* rhs_quad, int on_device);
void launch_traces(d4est_hip_plan* plan, const double* u, double* trace, bool ghost, const int* elist = nullptr, int n_list = 0, int parts = 3 /* see launch_flux */);
// Chebyshev update carried by the flux kernel's epilogue (flux_wave_kernel<true>): r = alpha (rhs - Au), p = r + beta p, u += p
struct ChebyFuse {
  const double* rhs = nullptr;
  double* p = nullptr;
  double* u = nullptr;
  double* u_out = nullptr;   // direct face kernel only: where the new iterate goes (it reads the neighbours' u)
  int skip_Au_store = 0;     // whole-operator kernel only: A u is consumed by the update and not stored (every iteration but the last)
  double* r = nullptr;
  double alpha = 0.0, beta = 0.0;
};
bool flux_can_fuse_update(d4est_hip_plan* plan);

// d4est_hip_direct.hip: the face terms straight from u (no trace arrays) on conforming uniform-degree plans
struct DirectFuse {   // Chebyshev update in the epilogue; the new iterate goes to u_out (the kernel reads the neighbours' u)
  const double* rhs = nullptr;
  double* p = nullptr;
  double* u_out = nullptr;
  double* r = nullptr;
  double alpha = 0.0, beta = 0.0;
  int skip_Au_store = 0;   // whole-operator form only: A u feeds the update in registers and is not written
  int pad = 0;
};
void direct_setup(d4est_hip_plan* plan, int N, int NQ, int ns0, int ns_stride, const double* C, const double* CD, const double* E);
void direct_destroy(d4est_hip_plan* plan);
bool direct_active(const d4est_hip_plan* plan);
double* direct_second_vector(d4est_hip_plan* plan);
// the direct kernel works on the listed elements only (the others are still read as neighbours); nullptr: every element.  The list
// (device ints) stays the caller's.  Only while direct_active(plan).
void direct_set_element_list(d4est_hip_plan* plan, const int* list_dev, int n_list);
bool direct_has_element_list(const d4est_hip_plan* plan);
bool direct_fused_ok(const d4est_hip_plan* plan);   // the volume term can ride in the same kernel (N = NQ in {6, 8}, one bucket ...)
// the elements with at least one ghost (+) side / with none (device lists; multi-rank plans: boundary traces feed the exchange,
// the interior elements' operator kernel runs while it is in flight)
void direct_ghost_split(d4est_hip_plan* plan, const int** bnd_list, int* n_bnd, const int** int_list, int* n_int);
void launch_direct_faces(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                         const double* robin_c, const double* robin_r, int vol_term);
// the same through the plan's face data (Robin arrays): vol_term = 0: Au += face terms of u; 1: Au = (volume + face terms) of u
void launch_flux_direct(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf = nullptr,
                        int vol_term = 0);
// elist / n_list: only these elements' face terms (the hybrid operator's dirty elements); the hanging-side record kernels keep their own list
// parts (locally refined plans under the hp split): 1 the conforming kernel only, 2 the record kernel only, 3 both
void launch_flux(d4est_hip_plan* plan, const double* trace, const double* ghost_trace, double* Au, const ChebyFuse* cf = nullptr,
                 const int* elist = nullptr, int n_list = 0, int parts = 3);
// the hybrid operator (d4est_hip_direct.hip): clean elements (all six sides conforming against a local element of the same degree, or
// the boundary) through the one-kernel path of their degree bucket, the rest through the two-phase kernels on lists
bool hybrid_pair_built(int N, int NQ);
// hanging-aware form (locally refined plans under the hp split): what a clean element's hanging side is to the direct kernel --
// kind 3: served by the mortar-record kernels (no contribution); kind 2: a small side, (+) block at goff in the plan's trace array, own
// block exported to export_off, combined factors at geom; kind < 0: an ordinary side
struct HybridSideOverride { int kind; long long goff; int export_off; int geom; };
void hybrid_setup(d4est_hip_plan* plan, const std::vector<char>& clean, const std::vector<const double*>& C, const std::vector<const double*>& CD,
                  const std::vector<const double*>& E, const std::vector<HybridSideOverride>* ov = nullptr, const char* form = "hanging-aware");
bool hybrid_hanging(const d4est_hip_plan* plan);
void hybrid_host_lists(const d4est_hip_plan* plan, const std::vector<int>** dirty, const std::vector<int>** ring);   // host copies of hybrid_lists   // the clean kernels read / write the trace array: record traces before, record flux after them
void hybrid_destroy(d4est_hip_plan* plan);
bool hybrid_active(const d4est_hip_plan* plan);
const char* hybrid_path(const d4est_hip_plan* plan);
void hybrid_lists(const d4est_hip_plan* plan, const int** dirty, int* n_dirty, const int** ring, int* n_ring);
void launch_hybrid_clean(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const double* robin_c, const double* robin_r,
                         int phase, const DirectFuse* cf = nullptr);
bool hybrid_can_fuse_update(const d4est_hip_plan* plan);   // the Chebyshev update can ride in the hybrid operator's kernels (hanging-aware form, all clean)
double* hybrid_second_vector(d4est_hip_plan* plan);
// the record flux kernel of an hp-split plan alone, optionally with the Chebyshev update of the elements it serves in its epilogue
void launch_flux_units(d4est_hip_plan* plan, const double* trace, const double* ghost_trace, double* Au, const ChebyFuse* cf);
bool faces_have_units(d4est_hip_plan* plan);
bool faces_hp(d4est_hip_plan* plan);         // the plan has hanging faces (mortar records)
bool faces_hp_split(d4est_hip_plan* plan);   // the plan has record kernels beside the conforming ones (launch_traces / launch_flux take `parts`)   // hp split with the unit record kernels (the form that can carry the update)
void launch_hybrid_dirty_stiffness(d4est_hip_plan* plan, const double* u, double* Au);
void launch_flux_hybrid_clean(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, int phase, const DirectFuse* cf = nullptr);   // (faces.hip: supplies the Robin arrays; phase 0 fork + launches, 1 join)
void faces_destroy(d4est_hip_plan* plan);

// d4est_hip_solver.hip
// traces, volume term, (exchange), flux; lhs_term: also the optional zeroth-order term of plan_set_lhs_coefficient
void apply_operator(d4est_hip_plan* plan, const double* u, double* Au, const ChebyFuse* cf = nullptr, bool lhs_term = true);
void launch_residual(d4est_hip_plan* plan, int n, const double* rhs, const double* Au, double* r);   // r = rhs - Au
void launch_residual_inplace(d4est_hip_plan* plan, int n, const double* rhs, double* r);              // r = rhs - r
void launch_residual_inplace_sub(d4est_hip_plan* plan, int n, const double* a, double* r);            // r = r - a
void ensure_solver_workspace(d4est_hip_plan* plan);
void add_lhs_mass_term(d4est_hip_plan* plan, const double* u, double* Au);   // Au += the plan's zeroth-order term, whichever form is set
// d4est_hip_mgmatrix.hip: the term as dense element blocks / as a Galerkin chain (multigrid matrix operator)
void add_lhs_blocks_term(d4est_hip_plan* plan, const double* u, double* Au);
void add_lhs_chain_term(d4est_hip_plan* plan, const double* u, double* Au);
void lhs_chain_destroy(d4est_hip_plan* plan);
// the term has a form that the fused operator kernels do not carry (blocks or chain): the volume kernel runs on its own and the term
// is added before the flux kernel
inline bool lhs_extra_term(const d4est_hip_plan* plan) { return plan->d_lhs_blocks != nullptr || plan->lhs_chain != nullptr; }
const double* ensure_lhs_wjc(d4est_hip_plan* plan);   // w J c at the quadrature nodes (formed on first use after the coefficient / geometry changed)
void launch_copy_blocks(hipStream_t stream, int n_blocks, const double* src, const long long* src_off, double* dst,
                        const long long* dst_off, const int* len);
void launch_dot(d4est_hip_plan* plan, int n, const double* x, const double* y, double* out_dev);
void launch_cheby_update(d4est_hip_plan* plan, int n, const double* rhs, const double* Au, double alpha, double beta, double* r,
                         double* p, double* u);
void cheby_iterate(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, double* r, int iter, double lmin, double lmax,
                   int compute_residual_at_end);
double cg_eigs(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, int imax, int use_new, double* hist_out);


}  // namespace d4est_hip
