// Shared declarations of the direct (trace-free) face kernels: d4est_hip_direct.hip (one wavefront per element, deg_quad <= 7) and
// d4est_hip_direct_mw.hip (one multi-wave workgroup per element, deg = deg_quad = 8 ... 15).
#pragma once
#include "d4est_hip_internal.h"

namespace d4est_hip {

struct DirectSide {
  int kcf;             // kind | code << 2 | fp << 5:  kind 0 boundary, 1 interface with a local (+) element, 2 with a ghost (+) element;
                       // code = flip0 | flip1<<1 | transpose<<2 applied when reading the (+) side; fp = face of the (+) element
  int nbr_ns;          // nodal offset of the (+) element (kind 1)
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
  bool mw = false;               // N > 8: the multi-wave kernel of d4est_hip_direct_mw.hip serves the plan
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
};


// d4est_hip_direct_mw.hip
bool direct_mw_built(int N, int NQ);
void launch_direct_mw(d4est_hip_plan* plan, DirectHost* dh, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                      const double* robin_c, const double* robin_r, int vmode, const DirectVol& vol, int n, int chunk);

}  // namespace d4est_hip
