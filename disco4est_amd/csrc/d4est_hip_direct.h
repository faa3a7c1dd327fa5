// Shared declarations of the direct (trace-free) face kernels: d4est_hip_direct.hip (one wavefront per element, deg_quad <= 7) and
// d4est_hip_direct_mw.hip (one multi-wave workgroup per element, deg = deg_quad = 8 ... 15).
#pragma once
#include "d4est_hip_internal.h"

namespace d4est_hip {

struct DirectSide {
  int kcf;             // kind | code << 2 | fp << 5:  kind 0 boundary, 1 interface with a local (+) element, 2 with a ghost (+) element
                       // (or, hanging-aware form, a small hanging side: the (+) block in the trace array), 3 (hanging-aware form) served elsewhere;
                       // code = flip0 | flip1<<1 | transpose<<2 applied when reading the (+) side; fp = face of the (+) element
  int nbr_ns;          // nodal offset of the (+) element (kind 1); hanging-aware form, kind 2: where the side's own mortar-node block is
                       // exported to in the trace array (-1: no export -- mixed-aware sides, whose block the ring's trace kernel writes)
  int geom;            // scalar offset of the side's mortar data (7 combined factors at 7*geom; Dirichlet / Robin data at geom)
  int pad;
};
// kind 2 only: offset of the (+) block in the ghost trace buffer, in its own array (read inside the ghost branch)
typedef long long DirectGhostOff;

struct DirectHost {
  int N = 0, NQ = 0;
  bool eo = false;
  int ns0 = 0, ns_stride = 0;
  DirectSide* d_sides = nullptr;
  DirectGhostOff* d_ghost_off = nullptr;
  double* d_ops = nullptr;   // C, CD, E, D^T E (plain: transposed; eo: even-odd tables), then rows 0 and N-1 of D
  double* d_u2 = nullptr;    // second solution vector of the fused Chebyshev update (see cheby_iterate_body)
  const int* d_list = nullptr;   // optional list of the elements the kernel works on (not owned; direct_set_element_list)
  int n_list = 0;
  mutable int order_ok = -1;     // 1: the plan's single bucket lists the elements in order (cached by direct_fused_ok)
  int hy_qs0 = 0, hy_qs_stride = -1;   // hybrid operator, one bucket = the whole plan in order: affine quadrature offsets (else by element id)
  bool hang = false;             // hybrid operator on a locally refined plan: sides of kind 3 / exports exist (VOL & 16 instances)
  bool mw = false;               // N > 8: the multi-wave kernel of d4est_hip_direct_mw.hip serves the plan
  int* d_bnd_list = nullptr;     // elements with a ghost (+) side, and the others (multi-rank plans; direct_ghost_split)
  int* d_int_list = nullptr;
  int n_bnd = 0, n_int = 0;
};

// VOL: the volume (stiffness) term of the element is applied by the same wavefront after its face terms and A u is written once --
// one kernel for the whole operator (u in, A u out: no read-modify-write of A u, one launch).  The face result waits in 8 registers
// per lane, in the layout of the volume kernel's coalesced store.
struct DirectVol {
  const double* metric = nullptr;    // 6 combined metric entries per quadrature node, element-blocked (plan->d_metric)
  const double* EBf = nullptr;       // even-odd tables of the volume operators (Bucket::d_EBf ...)
  const double* EGf = nullptr;
  const double* EBb = nullptr;
  const double* EGb = nullptr;
  const double* affine = nullptr;    // AFF: 6 numbers per element
  const double* wq = nullptr;        // AFF: quadrature weights
  int qs0 = 0, qs_stride = 0;
  const int* qs_list = nullptr;      // quadrature offset per element where the offsets are not affine (qs_stride < 0): a Schwarz
                                     // subdomain plan, whose element copies alias the mesh's metric
  const double* cq = nullptr;        // zeroth-order term (VOL & 4): w J c at the quadrature nodes (ensure_lhs_wjc)
  const double* EDq = nullptr;       // multi-wave kernel: even-odd tables of the differentiation matrix on the quadrature nodes and of
  const double* EDqT = nullptr;      // its transpose (collocated-gradient form of the volume term, stiffness_mw_element_cg)
  int stream = 0;                    // multi-wave kernel: stream mode (plan->stream_mode; with_ld, d4est_hip_wave.h)
};


// The direct kernels are short of scalar registers (operator rows travel through them): arguments that are needed late -- the smoother's
// vectors and coefficients, the volume term's tables, the side data pointers -- are NOT referenced as parameters (the compiler loads
// every referenced parameter at entry and then spills it through v_writelane / v_readlane for the whole kernel) but read from the
// kernel-argument segment where they are used.  DirectKernargs mirrors the parameter list: explicit arguments sit in the segment in
// order at their natural alignment, i.e. exactly as the members of this struct (checked against the code object's .args offsets by
// tests/test_capi.py::test_direct_kernarg_layout).
struct DirectKernargs {
  const double* u; const double* ghost_qtrace; double* Au; const DirectSide* sides; const DirectGhostOff* ghost_off;
  const double* ops; const double* geom; const double* bndry_q; const double* robin_c; const double* robin_r;
  int n_elem, ns0, ns_stride, xcd_chunk;
  DirectFuse cf;
  DirectVol vol;
  const int* elem_list;
};
typedef const DirectKernargs __attribute__((address_space(4))) * direct_kargs_ptr;
__device__ __forceinline__ direct_kargs_ptr direct_kargs() {
  unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(v));   // one opaque pointer per use site: the loads below it cannot be hoisted to the kernel entry
  return (direct_kargs_ptr)v;
}

__device__ __forceinline__ DirectFuse direct_load_fuse(direct_kargs_ptr K) {
  DirectFuse c;
  c.rhs = K->cf.rhs; c.p = K->cf.p; c.u_out = K->cf.u_out; c.r = K->cf.r;
  c.alpha = K->cf.alpha; c.beta = K->cf.beta; c.skip_Au_store = K->cf.skip_Au_store;
  return c;
}
__device__ __forceinline__ DirectVol direct_load_vol(direct_kargs_ptr K) {
  DirectVol v;
  v.metric = K->vol.metric; v.EBf = K->vol.EBf; v.EGf = K->vol.EGf; v.EBb = K->vol.EBb; v.EGb = K->vol.EGb;
  v.affine = K->vol.affine; v.wq = K->vol.wq; v.qs0 = K->vol.qs0; v.qs_stride = K->vol.qs_stride; v.qs_list = K->vol.qs_list;
  v.cq = K->vol.cq; v.EDq = K->vol.EDq; v.EDqT = K->vol.EDqT; v.stream = K->vol.stream;
  return v;
}

// d4est_hip_direct_mw.hip
bool direct_mw_built(int N, int NQ);
// vmode: 0 the face terms only (Au += ...), 1 / 2 the whole operator with the streamed / affine metric, + 4 with the zeroth-order term
void launch_direct_mw(d4est_hip_plan* plan, DirectHost* dh, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                      const double* robin_c, const double* robin_r, int vmode, const DirectVol& vol, int n, int chunk);

}  // namespace d4est_hip
