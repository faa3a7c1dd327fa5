// Face (mortar) part of the weak Laplacian: SIPG flux on conforming mortars and Dirichlet boundaries.
//
// Replaces, for all local (element, face) sides at once,
//   d4est_laplacian_compute_dudr              (src/dGMath/d4est_laplacian.c:237-282)  -> only face TRACES are formed
//   d4est_laplacian_flux_interface/_boundary  (src/dGMath/d4est_laplacian_flux.c:232-1014, :23-230)
//   d4est_laplacian_flux_sipg_interface/_dirichlet (src/dGMath/d4est_laplacian_flux_sipg.c:494-942, :15-336)
//   d4est_mortars_compute_flux_on_local_elements   (src/Mesh/d4est_mortars.c:601-840; the serial p4est_iterate walk)
//
// Design: element-centric and two-phase.
//  (1) trace kernel: every element writes, for each of its six sides, the traces of u and of du/dr_{0,1,2}
//      ALREADY INTERPOLATED to the side's mortar quadrature nodes (4 T doubles per side, T = (deg_mortar_quad+1)^2).
//      The reference interpolates every side twice per apply -- once as the (-) side of its own flux call and once as
//      the (+) side of the neighbour's -- here it happens once, and the mortar-node trace is also the only thing a
//      neighbouring GPU needs (instead of the reference's whole ghost elements).
//  (2) flux kernel: one workgroup per element walks its six sides, reads its own and the neighbour's mortar-node
//      traces (the neighbour's re-ordered by the p4est flip/transpose code), evaluates the three SIPG terms,
//      integrates and projects back (one tensor apply), scatters the four lifted fields into LDS volume fields and
//      applies  W_0 + sum_l D_l^T W_l  into Au_e once: race-free and deterministic, no atomics.
// The reference's 24 doubles of mortar geometry per quadrature node are pre-combined at set-up into 7:
//   am_i = sum_d sj n_d (dr_i/dx_d)^-,  ap_i = sum_d sj n_d (dr_i/dx_d)^+ (re-ordered to the (-) side),  s3 = sj sigma.
#include <algorithm>
#include <map>
#include <tuple>

#include "d4est_hip_internal.h"
#include "d4est_hip_topology.h"
#include "d4est_hip_maps.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_wave.h"

namespace d4est_hip {

struct SideDesc {
  int kind;           // 0 boundary, 1 interface with local (+), 2 interface with ghost (+)
  int code;           // flip0 | flip1<<1 | transpose<<2   (dGMath/d4est_operators.c:2031-2081)
  int NQ;             // mortar quadrature nodes/dir
  int offC;           // (NQ x N)  side nodes -> mortar quadrature nodes (p-prolong then Lobatto->quadrature)
  int offCD;          // (NQ x N)  C * D: tangential derivative fused with the interpolation (fast trace kernel)
  int pad0;
  int offE;           // (N x NQ)  mortar quadrature -> side operator (P^T I^T W)
  int geom;           // scalar stride S of the side (face_geom at 7*S; Dirichlet data at S)
  long long qoff;     // offset of this side's mortar-node trace block (4 T doubles) in the local buffer
  long long nbr_qoff; // offset of the (+) side's block (local buffer for kind 1, ghost buffer for kind 2)
};

struct ElemDesc {
  int N;     // nodes per direction
  int ns;    // nodal stride
  int offD;  // offset of the zero-padded 8 x 8 (fast path) or N x N (generic) derivative matrix inside face_ops
  int pad;
};

// uniform plans (one degree, one mortar degree, contiguous strides): the trace kernel computes its addresses instead of loading
// descriptors, which removes one dependent global-load latency from every workgroup (the kernel is latency bound)
struct TraceUniform {
  int N, NQ, offC, offCD, offD, ns0, ns_stride, pad;  // N == 0: not uniform
  long long q0, q_stride;                              // qoff of side s = q0 + s * q_stride
};

struct GhostSideDesc {
  int N;       // nodes/dir of the ghost element
  int f;       // its face
  int NQ;
  int offC;    // (NQ x N)
  int offD;    // N x N
  int pad;
  long long u_off;  // offset of the ghost element in the packed ghost vector
  long long goff;   // offset of the block in the ghost trace buffer
};


// volume index of face node (a,b) of face f (a fastest; tangential axes in increasing order)
__device__ inline int face_vol_index(int f, int N, int a, int b) {
  const int dir = f >> 1, fix = face_fix(f, N);
  if (dir == 0) return fix + N * (a + N * b);
  if (dir == 1) return a + N * (fix + N * b);
  return a + N * (b + N * fix);
}

// value c (0: u, 1..3: du/dr_{c-1}) at face node (a,b) of face f from the element values ue (N^3, x fastest)
__device__ inline double nodal_trace(const double* ue, const double* D /* row-major, leading dim ldD */, int ldD, int N, int f,
                                     int a, int b, int c) {
  const int v = face_vol_index(f, N, a, b);
  if (c == 0) return ue[v];
  const int d = c - 1;
  const int stride = (d == 0) ? 1 : (d == 1 ? N : N * N);
  const int pos = (v / stride) % N;
  const int base = v - pos * stride;
  double val = 0.0;
  for (int i = 0; i < N; ++i) val = fma(D[pos * ldD + i], ue[base + i * stride], val);
  return val;
}

// ---------------------------------------------------------------------------
// generic tensor apply (runtime sizes, whole workgroup): out (rows x rows) = (op (x) op) in (cols x cols)
// ---------------------------------------------------------------------------
__device__ inline void apply2d(const double* __restrict__ op, int rows, int cols, const double* in, double* tmp, double* out,
                               int nfields, int in_stride, int out_stride, int tmp_stride) {
  for (int idx = threadIdx.x; idx < nfields * rows * cols; idx += blockDim.x) {
    const int fld = idx / (rows * cols), r = idx % (rows * cols), ap = r % rows, b = r / rows;
    const double* x = in + fld * in_stride + cols * b;
    double s = 0.0;
    for (int a = 0; a < cols; ++a) s = fma(op[ap * cols + a], x[a], s);
    tmp[fld * tmp_stride + ap + rows * b] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nfields * rows * rows; idx += blockDim.x) {
    const int fld = idx / (rows * rows), r = idx % (rows * rows), ap = r % rows, bp = r / rows;
    const double* x = tmp + fld * tmp_stride + ap;
    double s = 0.0;
    for (int b = 0; b < cols; ++b) s = fma(op[bp * cols + b], x[rows * b], s);
    out[fld * out_stride + ap + rows * bp] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// (1) generic trace kernel: any degree, one 256-thread workgroup per element
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_generic_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                            const SideDesc* __restrict__ sd, const ElemDesc* __restrict__ ed,
                                                            const double* __restrict__ face_ops, int n_elem, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;                     // 4 nodal fields
  double* tmp = A + 4 * fld_stride;
  double* Q = tmp + 4 * fld_stride;     // 4 fields at the mortar nodes
  double* ue = Q + 4 * fld_stride;
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    double* Ds = ue + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) ue[i] = u[el.ns + i];
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = face_ops[el.offD + i];
    __syncthreads();
    for (int f = 0; f < 6; ++f) {
      const SideDesc d = sd[6 * e + f];
      const int NQ = d.NQ, T = NQ * NQ;
      for (int idx = threadIdx.x; idx < 4 * N2; idx += blockDim.x) {
        const int c = idx / N2, ab = idx % N2;
        A[c * fld_stride + ab] = nodal_trace(ue, Ds, N, N, f, ab % N, ab / N, c);
      }
      __syncthreads();
      apply2d(face_ops + d.offC, NQ, N, A, tmp, Q, 4, fld_stride, fld_stride, fld_stride);
      for (int idx = threadIdx.x; idx < 4 * T; idx += blockDim.x) qtrace[d.qoff + idx] = Q[(idx / T) * fld_stride + idx % T];
      __syncthreads();
    }
  }
}

// ghost sides: the same for face f of a ghost element given as a whole element (reference-style ghost data)
__global__ __launch_bounds__(256) void ghost_trace_kernel(const double* __restrict__ u_ghost, double* __restrict__ ghost_qtrace,
                                                          const GhostSideDesc* __restrict__ gd, const double* __restrict__ face_ops,
                                                          int n_ghost_sides, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;
  double* tmp = A + 4 * fld_stride;
  double* Q = tmp + 4 * fld_stride;
  double* ue = Q + 4 * fld_stride;
  for (int s = blockIdx.x; s < n_ghost_sides; s += gridDim.x) {
    const GhostSideDesc g = gd[s];
    const int N = g.N, N2 = N * N, N3 = N2 * N, T = g.NQ * g.NQ;
    double* Ds = ue + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) ue[i] = u_ghost[g.u_off + i];
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = face_ops[g.offD + i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * N2; idx += blockDim.x) {
      const int c = idx / N2, ab = idx % N2;
      A[c * fld_stride + ab] = nodal_trace(ue, Ds, N, N, g.f, ab % N, ab / N, c);
    }
    __syncthreads();
    apply2d(face_ops + g.offC, g.NQ, N, A, tmp, Q, 4, fld_stride, fld_stride, fld_stride);
    for (int idx = threadIdx.x; idx < 4 * T; idx += blockDim.x) ghost_qtrace[g.goff + idx] = Q[(idx / T) * fld_stride + idx % T];
    __syncthreads();
  }
}

// Dirichlet data: Lobatto face nodes -> mortar quadrature nodes (d4est_laplacian_flux_sipg.c:80-107), set-up time
__global__ __launch_bounds__(64) void bndry_interp_kernel(const double* __restrict__ g_lobatto, double* __restrict__ g_quad,
                                                          const SideDesc* __restrict__ sd, const ElemDesc* __restrict__ ed,
                                                          const int* __restrict__ side_bndry_stride,
                                                          const double* __restrict__ face_ops, int n_sides) {
  for (int s = blockIdx.x; s < n_sides; s += gridDim.x) {
    const SideDesc d = sd[s];
    if (d.kind != 0) continue;
    const int N = ed[s / 6].N, NQ = d.NQ;
    const double* C = face_ops + d.offC;
    const double* g = g_lobatto + side_bndry_stride[s];
    for (int k = threadIdx.x; k < NQ * NQ; k += blockDim.x) {
      const int ap = k % NQ, bp = k / NQ;
      double v = 0.0;
      for (int b = 0; b < N; ++b) {
        double r = 0.0;
        for (int a = 0; a < N; ++a) r = fma(C[ap * N + a], g[a + N * b], r);
        v = fma(C[bp * N + b], r, v);
      }
      g_quad[d.geom + k] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// set-up: 7 combined geometric factors per mortar quadrature node
// ---------------------------------------------------------------------------
__device__ inline double sipg_penalty(int fcn, int deg_m, double h_m, int deg_p, double h_p, double prefactor) {
  // src/dGMath/d4est_laplacian_flux_sipg.c:945-1005
  if (fcn == 0) {
    const double max_deg = (deg_m > deg_p) ? deg_m : deg_p, min_h = (h_m < h_p) ? h_m : h_p;
    return (prefactor * max_deg * max_deg) / min_h;
  } else if (fcn == 1) {
    const double mean_p = .5 * (deg_m + deg_p), mean_h = .5 * (h_m + h_p);
    return (prefactor * mean_p * mean_p) / mean_h;
  } else if (fcn == 2) {
    const double max_deg = (deg_m > deg_p) ? deg_m : deg_p, min_h = (h_m < h_p) ? h_m : h_p;
    return (prefactor * (max_deg + 1) * (max_deg + 1)) / min_h;
  }
  return prefactor * .5 * (deg_m * deg_m / h_m + deg_p * deg_p / h_p);
}

__global__ __launch_bounds__(64) void face_geom_kernel(const SideDesc* __restrict__ sd, const int* __restrict__ side_deg_m,
                                                       const int* __restrict__ side_deg_p, int n_sides,
                                                       const double* __restrict__ sj, const double* __restrict__ nrm,
                                                       const double* __restrict__ drst_m, const double* __restrict__ drst_p,
                                                       const double* __restrict__ hm, const double* __restrict__ hp,
                                                       double prefactor, int fcn, double* __restrict__ geom) {
  for (int s = blockIdx.x; s < n_sides; s += gridDim.x) {
    const SideDesc d = sd[s];
    const int NQ = d.NQ, T = NQ * NQ;
    const size_t S = (size_t)d.geom;
    for (int k = threadIdx.x; k < T; k += blockDim.x) {
      const int a = k % NQ, b = k / NQ;
      const int kp = (d.kind == 0) ? k : reorder_index(d.code, NQ - 1, a, b);
      const double sjk = sj[S + k];
      double sn[3];
      for (int x = 0; x < 3; ++x) sn[x] = sjk * nrm[3 * S + (size_t)x * T + k];
      for (int i = 0; i < 3; ++i) {
        double am = 0.0, ap = 0.0;
        for (int x = 0; x < 3; ++x) {
          am += sn[x] * drst_m[9 * S + (size_t)(i + 3 * x) * T + k];
          if (d.kind != 0) ap += sn[x] * drst_p[9 * S + (size_t)(i + 3 * x) * T + kp];
        }
        geom[7 * S + (size_t)i * T + k] = am;
        geom[7 * S + (size_t)(3 + i) * T + k] = ap;
      }
      const int dm = side_deg_m[s], dp = (d.kind == 0) ? dm : side_deg_p[s];
      const double hpk = (d.kind == 0) ? hm[S + k] : hp[S + k];
      geom[7 * S + (size_t)6 * T + k] = sjk * sipg_penalty(fcn, dm, hm[S + k], dp, hpk, prefactor);
    }
  }
}

// ---------------------------------------------------------------------------
// (2) generic flux kernel: any degree, one 256-thread workgroup per element, sides in sequence
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flux_generic_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                           double* __restrict__ Au, const SideDesc* __restrict__ sd,
                                                           const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                           const double* __restrict__ geom, const double* __restrict__ bndry_q,
                                                           const double* __restrict__ robin_c, const double* __restrict__ robin_r,
                                                           int n_elem, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;                     // 4 term fields at the mortar nodes
  double* tmp = A + 4 * fld_stride;
  double* R = tmp + 4 * fld_stride;     // 4 fields on the side's nodes
  double* acc = R + 4 * fld_stride;
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    double* Ds = acc + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) acc[i] = 0.0;
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = face_ops[el.offD + i];
    __syncthreads();
    for (int f = 0; f < 6; ++f) {
      const SideDesc d = sd[6 * e + f];
      const int NQ = d.NQ, T = NQ * NQ;
      const double* g = geom + (size_t)7 * d.geom;
      const double* qm = qtrace + d.qoff;
      const double* qp = ((d.kind == 2) ? ghost_qtrace : qtrace) + d.nbr_qoff;
      for (int k = threadIdx.x; k < T; k += blockDim.x) {
        if (d.kind == 0 && robin_c) {
          // Robin boundary (d4est_laplacian_flux_sipg.c:339-489): only  sj (coeff u_m - rhs), integrated and lifted
          A[k] = robin_c[d.geom + k] * qm[k] - robin_r[d.geom + k];
          for (int l = 0; l < 3; ++l) A[(1 + l) * fld_stride + k] = 0.0;
          continue;
        }
        const int kp = (d.kind == 0) ? k : reorder_index(d.code, NQ - 1, k % NQ, k / NQ);
        const double um = qm[k];
        const double up = (d.kind == 0) ? bndry_q[d.geom + k] : qp[kp];
        double t1 = 0.0, am[3];
        for (int i = 0; i < 3; ++i) {
          am[i] = g[i * T + k];
          t1 += am[i] * qm[(1 + i) * T + k];
          if (d.kind != 0) t1 += g[(3 + i) * T + k] * qp[(1 + i) * T + kp];
        }
        const double jump = um - up;
        // interface: t1 = -1/2 sj n.(grad u_m + grad u_p), t2_l = -1/2 am_l [u]; boundary: t1 = -sj n.grad u_m, t2_l = -am_l (u - g)
        const double w1 = (d.kind != 0) ? -0.5 : -1.0;
        A[k] = w1 * t1 + g[6 * T + k] * jump;
        for (int l = 0; l < 3; ++l) A[(1 + l) * fld_stride + k] = w1 * am[l] * jump;
      }
      __syncthreads();
      apply2d(face_ops + d.offE, N, NQ, A, tmp, R, 4, fld_stride, fld_stride, fld_stride);
      // lift (+ D_l^T for the term-2 fields) into the element accumulator
      const int dir = f >> 1, fix = face_fix(f, N);
      const int sdir = (dir == 0) ? 1 : (dir == 1 ? N : N2);
      for (int idx = threadIdx.x; idx < N3; idx += blockDim.x) {
        const int pos = (idx / sdir) % N;
        int a, b;
        if (dir == 0) { a = (idx / N) % N; b = idx / N2; }
        else if (dir == 1) { a = idx % N; b = idx / N2; }
        else { a = idx % N; b = (idx / N) % N; }
        double v = Ds[fix * N + pos] * R[(1 + dir) * fld_stride + a + N * b];
        if (pos == fix) {
          v += R[a + N * b];
          const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;  // tangential reference directions of a and b
          double s0 = 0.0, s1 = 0.0;
          for (int q = 0; q < N; ++q) {
            s0 = fma(Ds[q * N + a], R[(1 + t0) * fld_stride + q + N * b], s0);
            s1 = fma(Ds[q * N + b], R[(1 + t1d) * fld_stride + a + N * q], s1);
          }
          v += s0 + s1;
        }
        acc[idx] += v;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < N3; i += blockDim.x) Au[el.ns + i] += acc[i];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Non-conforming (hanging 1 <-> 4) meshes: mortar-record path.  Every side owns 1 record (conforming, boundary, or the
// hanging "small" side: faces_m = 4, faces_p = 1, seen from one of the four small elements) or 4 records (the "big" side:
// faces_m = 1, faces_p = 4).  A record's mortar-node trace block is produced by its own element with
//   conforming / small:  p-prolong + interpolation              (Mesh/d4est_mortars.c:556-566, d4est_laplacian_flux.c:659-694)
//   big, sub-mortar i:   hp-prolong child (i & 1, i >> 1) + interpolation   (Mesh/d4est_mortars.c:568-579)
// and consumed by the element itself and by the element across the mortar.  du/dr in a block is with respect to the
// producing element's own reference coordinates; the mortar's metric (given on the mortar-sized cell) is twice the big
// element's, hence the factors fm / fp = 1/2 on the big element's gradient and w2 = 1/2 on the big side's term 2
// (d4est_laplacian_flux.c:905-915, d4est_laplacian_flux_sipg.c:806-808).
// ---------------------------------------------------------------------------
struct HpMortar {
  int elem, face;
  int kind;          // 0 boundary, 1 interface with a local (+) element, 2 with a ghost (+) element
  int code;          // reorder code applied when reading the (+) block
  int N, NQ;         // side nodes / mortar quadrature nodes per direction
  int offCa, offCb;  // (NQ x N) side -> mortar quadrature nodes along the face axes a, b (hp_ops)
  int offCDa, offCDb; // C . D: tangential derivative fused with the interpolation (MFMA kernels)
  int offEa, offEb;  // (N x NQ) mortar quadrature nodes -> side
  int first, last;   // first / last record of its side
  int gidx;          // scalar index of the mortar's first node in the precombined geometry / boundary arrays (S + off)
  int u_shift;       // offset (doubles) from the (+) block to the block whose u field this mortar reads; 0 except on a small side whose
                     // face pair the reference re-orients non-geometrically (see faces_setup_hp)
  double fm, fp, w2; // hanging-face factors (set-up only: face_geom_hp_kernel folds them into the geometric factors; w2 = fm)
  double hang;       // side_hang of the record's side as a number (0 conforming, 1 big, 2 small): lets the tiled kernels serve the hanging sides only (hp split)
  long long qoff, nbr_qoff;
};

struct HpGeomSrc {   // where the record's factors sit in the reference-layout arrays (set-up only)
  int S, off, Ttot, off_p;
  int deg_m, deg_p, pad0, pad1;
};

// out (rows x rows) = (opb (x) opa) in (cols x cols): opa contracts the fast face index a, opb the slow index b
__device__ inline void apply2d_ab(const double* __restrict__ opa, const double* __restrict__ opb, int rows, int cols, const double* in,
                                  double* tmp, double* out, int nfields, int in_stride, int out_stride, int tmp_stride, bool accumulate) {
  for (int idx = threadIdx.x; idx < nfields * rows * cols; idx += blockDim.x) {
    const int fld = idx / (rows * cols), r = idx % (rows * cols), ap = r % rows, b = r / rows;
    const double* x = in + fld * in_stride + cols * b;
    double s = 0.0;
    for (int a = 0; a < cols; ++a) s = fma(opa[ap * cols + a], x[a], s);
    tmp[fld * tmp_stride + ap + rows * b] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nfields * rows * rows; idx += blockDim.x) {
    const int fld = idx / (rows * rows), r = idx % (rows * rows), ap = r % rows, bp = r / rows;
    const double* x = tmp + fld * tmp_stride + ap;
    double s = 0.0;
    for (int b = 0; b < cols; ++b) s = fma(opb[bp * cols + b], x[rows * b], s);
    if (accumulate) out[fld * out_stride + ap + rows * bp] += s;
    else out[fld * out_stride + ap + rows * bp] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void trace_hp_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                       const HpMortar* __restrict__ md, const int* __restrict__ elem_first,
                                                       const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                       const double* __restrict__ hp_ops, int n_elem, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;                     // 4 nodal fields of the current side
  double* tmp = A + 4 * fld_stride;
  double* Q = tmp + 4 * fld_stride;     // 4 fields at the mortar nodes
  double* ue = Q + 4 * fld_stride;
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    double* Ds = ue + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) ue[i] = u[el.ns + i];
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = face_ops[el.offD + i];
    __syncthreads();
    for (int r = elem_first[e]; r < elem_first[e + 1]; ++r) {
      const HpMortar m = md[r];
      const int NQ = m.NQ, T = NQ * NQ;
      if (m.first) {
        for (int idx = threadIdx.x; idx < 4 * N2; idx += blockDim.x) {
          const int c = idx / N2, ab = idx % N2;
          A[c * fld_stride + ab] = nodal_trace(ue, Ds, N, N, m.face, ab % N, ab / N, c);
        }
        __syncthreads();
      }
      apply2d_ab(hp_ops + m.offCa, hp_ops + m.offCb, NQ, N, A, tmp, Q, 4, fld_stride, fld_stride, fld_stride, false);
      for (int idx = threadIdx.x; idx < 4 * T; idx += blockDim.x) qtrace[m.qoff + idx] = Q[(idx / T) * fld_stride + idx % T];
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(64) void face_geom_hp_kernel(const HpMortar* __restrict__ md, const HpGeomSrc* __restrict__ gs, int n_rec,
                                                          const double* __restrict__ sj, const double* __restrict__ nrm,
                                                          const double* __restrict__ drst_m, const double* __restrict__ drst_p,
                                                          const double* __restrict__ hm, const double* __restrict__ hp,
                                                          double prefactor, int fcn, double* __restrict__ geom) {
  for (int r = blockIdx.x; r < n_rec; r += gridDim.x) {
    const HpMortar m = md[r];
    const HpGeomSrc g = gs[r];
    const int NQ = m.NQ, T = NQ * NQ;
    const size_t S = (size_t)g.S, TT = (size_t)g.Ttot;
    for (int k = threadIdx.x; k < T; k += blockDim.x) {
      const int a = k % NQ, b = k / NQ;
      const int kp = (m.kind == 0) ? k : reorder_index(m.code, NQ - 1, a, b);
      const double sjk = sj[S + g.off + k];
      double sn[3];
      for (int x = 0; x < 3; ++x) sn[x] = sjk * nrm[3 * S + (size_t)x * TT + g.off + k];
      for (int i = 0; i < 3; ++i) {
        double am = 0.0, ap = 0.0;
        for (int x = 0; x < 3; ++x) {
          am += sn[x] * drst_m[9 * S + (size_t)(i + 3 * x) * TT + g.off + k];
          // (+) side factors are stored in the (+) side's sub-mortar order and orientation (d4est_laplacian_flux.c:858-900)
          if (m.kind != 0) ap += sn[x] * drst_p[9 * S + (size_t)(i + 3 * x) * TT + g.off_p + kp];
        }
        // the hanging-face factors of the reference (d4est_laplacian_flux.c:907-915: x 0.5 on the big element's gradient and term 2 on
        // its half-size mortars, x 0.5 on the big element's gradient seen from a small side) are folded in here -- exact scalings --,
        // so that a record's SIPG terms read like a conforming side's and a small side can go to the conforming flux kernel (hp split)
        geom[7 * (size_t)m.gidx + (size_t)i * T + k] = m.fm * am;
        geom[7 * (size_t)m.gidx + (size_t)(3 + i) * T + k] = m.fp * ap;
      }
      const double hmk = hm[S + g.off + k], hpk = (m.kind == 0) ? hmk : hp[S + g.off + k];
      geom[7 * (size_t)m.gidx + (size_t)6 * T + k] = sjk * sipg_penalty(fcn, g.deg_m, hmk, (m.kind == 0) ? g.deg_m : g.deg_p, hpk, prefactor);
    }
  }
}

__global__ __launch_bounds__(256) void flux_hp_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                      double* __restrict__ Au,
                                                      const HpMortar* __restrict__ md, const int* __restrict__ elem_first,
                                                      const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                      const double* __restrict__ hp_ops, const double* __restrict__ geom,
                                                      const double* __restrict__ bndry_q, const double* __restrict__ robin_c,
                                                      const double* __restrict__ robin_r, int n_elem, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;                     // 4 term fields at the mortar nodes
  double* tmp = A + 4 * fld_stride;
  double* R = tmp + 4 * fld_stride;     // 4 fields on the side's nodes, summed over the side's mortars
  double* acc = R + 4 * fld_stride;
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    double* Ds = acc + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) acc[i] = 0.0;
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = face_ops[el.offD + i];
    __syncthreads();
    for (int r = elem_first[e]; r < elem_first[e + 1]; ++r) {
      const HpMortar m = md[r];
      const int NQ = m.NQ, T = NQ * NQ, f = m.face;
      const double* g = geom + (size_t)7 * m.gidx;
      const double* qm = qtrace + m.qoff;
      const double* qp = ((m.kind == 2) ? ghost_qtrace : qtrace) + m.nbr_qoff;
      for (int k = threadIdx.x; k < T; k += blockDim.x) {
        if (m.kind == 0 && robin_c) {
          A[k] = robin_c[m.gidx + k] * qm[k] - robin_r[m.gidx + k];
          for (int l = 0; l < 3; ++l) A[(1 + l) * fld_stride + k] = 0.0;
          continue;
        }
        const int kp = (m.kind == 0) ? k : reorder_index(m.code, NQ - 1, k % NQ, k / NQ);
        const double um = qm[k];
        const double up = (m.kind == 0) ? bndry_q[m.gidx + k] : qp[kp + m.u_shift];
        double tm = 0.0, tp = 0.0, am[3];
        for (int i = 0; i < 3; ++i) {
          am[i] = g[i * T + k];
          tm += am[i] * qm[(1 + i) * T + k];
          if (m.kind != 0) tp += g[(3 + i) * T + k] * qp[(1 + i) * T + kp];
        }
        const double jump = um - up;
        const double w1 = (m.kind != 0) ? -0.5 : -1.0;
        A[k] = w1 * (tm + tp) + g[6 * T + k] * jump;   // (fm, fp, w2: folded into the factors by face_geom_hp_kernel)
        for (int l = 0; l < 3; ++l) A[(1 + l) * fld_stride + k] = w1 * am[l] * jump;
      }
      __syncthreads();
      apply2d_ab(hp_ops + m.offEa, hp_ops + m.offEb, N, NQ, A, tmp, R, 4, fld_stride, fld_stride, fld_stride, !m.first);
      if (!m.last) continue;
      const int dir = f >> 1, fix = face_fix(f, N);
      const int sdir = (dir == 0) ? 1 : (dir == 1 ? N : N2);
      for (int idx = threadIdx.x; idx < N3; idx += blockDim.x) {
        const int pos = (idx / sdir) % N;
        int a, b;
        if (dir == 0) { a = (idx / N) % N; b = idx / N2; }
        else if (dir == 1) { a = idx % N; b = idx / N2; }
        else { a = idx % N; b = (idx / N) % N; }
        double v = Ds[fix * N + pos] * R[(1 + dir) * fld_stride + a + N * b];
        if (pos == fix) {
          v += R[a + N * b];
          const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
          double s0 = 0.0, s1 = 0.0;
          for (int q = 0; q < N; ++q) {
            s0 = fma(Ds[q * N + a], R[(1 + t0) * fld_stride + q + N * b], s0);
            s1 = fma(Ds[q * N + b], R[(1 + t1d) * fld_stride + a + N * q], s1);
          }
          v += s0 + s1;
        }
        acc[idx] += v;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < N3; i += blockDim.x) Au[el.ns + i] += acc[i];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// fast path (every N, Np, NQ <= 8, i.e. p <= 7): workgroup of six wavefronts per element, wavefront f <-> side f,
// lane <-> node of a fixed 8 x 8 grid (index a + 8 b).  Operators are zero-padded to 8 x 8 in LDS so the tensor
// applies are branch-free and fully unrolled (the padding multiplies zeros).
// ---------------------------------------------------------------------------
constexpr int kFW = 8;

// out_c(a',b') = sum_{a,b} op[a'][a] op[b'][b] in_c(a,b); lane = a' + 8 b'; result in registers
// the buffer handed to wave_apply2d belongs to the calling WAVE only (each wave of flux_wave_kernel owns s_in[f]), so its hand-offs
// through LDS need a wave-level fence, not a workgroup barrier
__device__ __forceinline__ void wave_private_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// two-buffer form of wave_apply2d (first pass in -> tmp): one wave-level fence less per pass than the in-place form
template <int NF>
__device__ __forceinline__ void wave_apply2d_tmp(const double* op /*[8][8] padded*/, const double* in /*[NF][64]*/,
                                                 double* tmp /*[NF][64]*/, int lane, double* out /*[NF]*/) {
  const int lo = lane & 7, hi = lane >> 3;
  double c[kFW];
#pragma unroll
  for (int a = 0; a < kFW; ++a) c[a] = op[lo * 8 + a];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < kFW; ++a) s = fma(c[a], in[f * 64 + a + 8 * hi], s);
    tmp[f * 64 + lane] = s;
  }
  wave_private_lds_fence();
#pragma unroll
  for (int b = 0; b < kFW; ++b) c[b] = op[hi * 8 + b];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < kFW; ++b) s = fma(c[b], tmp[f * 64 + lo + 8 * b], s);
    out[f] = s;
  }
  wave_private_lds_fence();
}

template <int NF>
__device__ __forceinline__ void wave_apply2d(const double* op /*[8][8] padded*/, double* buf /*[NF][64], overwritten*/, int lane,
                                             double* out /*[NF]*/) {
  // both passes work in place: a wave executes in lockstep, so every lane has its row / column in registers before the
  // (fenced) stores of the pass overwrite the buffer
  const int lo = lane & 7, hi = lane >> 3;
  double c[kFW], t[NF];
#pragma unroll
  for (int a = 0; a < kFW; ++a) c[a] = op[lo * 8 + a];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < kFW; ++a) s = fma(c[a], buf[f * 64 + a + 8 * hi], s);
    t[f] = s;
  }
  wave_private_lds_fence();
#pragma unroll
  for (int f = 0; f < NF; ++f) buf[f * 64 + lane] = t[f];
  wave_private_lds_fence();
#pragma unroll
  for (int b = 0; b < kFW; ++b) c[b] = op[hi * 8 + b];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < kFW; ++b) s = fma(c[b], buf[f * 64 + lo + 8 * b], s);
    out[f] = s;
  }
  wave_private_lds_fence();
}

__global__ __launch_bounds__(384, 6) void trace_wave_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                         const SideDesc* __restrict__ sd, const ElemDesc* __restrict__ ed,
                                                         const double* __restrict__ face_ops, int n_elem, TraceUniform uni) {
  // Per side only TWO nodal fields are formed -- the trace tr(a,b) of u and the normal derivative n(a,b) -- because the
  // tangential derivatives commute with the interpolation:  (C (x) C)(D_a tr) = ((C D) (x) C) tr.  Pass 1 contracts the
  // face index a with C and CD, pass 2 the index b: 56 FMAs and ~90 LDS reads per lane instead of 152 / 130.
  __shared__ double s_u[512];
  __shared__ double s_D[64];          // zero-padded 8 x 8
  __shared__ double s_in[6][2][64];   // tr, n
  __shared__ double s_tmp[6][3][64];  // C tr, CD tr, C n
  __shared__ double s_C[6][2][64];    // C, CD zero-padded 8 x 8
  const int f = threadIdx.x >> 6, lane = threadIdx.x & 63, lo = lane & 7, hi = lane >> 3;
  const int dir = f >> 1;
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;  // reference directions of the face indices a and b
  int e = blockIdx.x;
  ElemDesc edn{};
  SideDesc dn{};
  auto uniform_desc = [&](int e_, ElemDesc& el_, SideDesc& d_) {
    el_.N = uni.N; el_.ns = uni.ns0 + e_ * uni.ns_stride; el_.offD = uni.offD;
    d_.NQ = uni.NQ; d_.offC = uni.offC; d_.offCD = uni.offCD; d_.qoff = uni.q0 + (long long)(6 * e_ + f) * uni.q_stride;
  };
  if (uni.N > 0) uniform_desc(e < n_elem ? e : 0, edn, dn);
  else {
    edn = ed[e < n_elem ? e : 0];
    dn = sd[6 * (e < n_elem ? e : 0) + f];
  }
  for (; e < n_elem; e += gridDim.x) {
    const ElemDesc el = edn;
    const SideDesc d = dn;
    {
      const int en = e + gridDim.x;
      if (en < n_elem) {
        if (uni.N > 0) uniform_desc(en, edn, dn);
        else {
          edn = ed[en];
          dn = sd[6 * en + f];
        }
      }
    }
    const int N = el.N, N2 = N * N, N3 = N2 * N, NQ = d.NQ, T = NQ * NQ;
    double uv0 = 0.0, uv1 = 0.0;
    if ((int)threadIdx.x < N3) uv0 = u[el.ns + threadIdx.x];
    if ((int)threadIdx.x + 384 < N3) uv1 = u[el.ns + threadIdx.x + 384];
    const bool opl = hi < NQ && lo < N;
    const double opc = opl ? face_ops[d.offC + hi * N + lo] : 0.0;
    const double opcd = opl ? face_ops[d.offCD + hi * N + lo] : 0.0;
    const double dval = (threadIdx.x < 64) ? face_ops[el.offD + threadIdx.x] : 0.0;
    if ((int)threadIdx.x < N3) s_u[threadIdx.x] = uv0;
    if ((int)threadIdx.x + 384 < N3) s_u[threadIdx.x + 384] = uv1;
    if (threadIdx.x < 64) s_D[threadIdx.x] = dval;
    s_C[f][0][lane] = opc;
    s_C[f][1][lane] = opcd;
    __syncthreads();
    // ---- nodal trace and normal derivative at lane (a = lo, b = hi)
    {
      const int sn = (dir == 0) ? 1 : (dir == 1 ? N : N2);
      const int sa = (dir == 0) ? N : 1, sb = (dir == 2) ? N : N2;
      const int fix = face_fix(f, N);
      double tr = 0.0, nd = 0.0;
      if (lo < N && hi < N) {
        const int base = lo * sa + hi * sb;
        tr = s_u[base + fix * sn];
#pragma unroll
        for (int i = 0; i < kFW; ++i) nd = fma(s_D[fix * 8 + i], s_u[base + (i < N ? i : N - 1) * sn], nd);  // padded D columns are 0
      }
      s_in[f][0][lane] = tr;
      s_in[f][1][lane] = nd;
    }
    __syncthreads();
    // ---- pass 1: lane (a' = lo, b = hi): contract the face index a
    {
      double c1[kFW], c2[kFW];
#pragma unroll
      for (int a = 0; a < kFW; ++a) { c1[a] = s_C[f][0][lo * 8 + a]; c2[a] = s_C[f][1][lo * 8 + a]; }
      double P = 0.0, R = 0.0, S = 0.0;
#pragma unroll
      for (int a = 0; a < kFW; ++a) {
        const double t = s_in[f][0][a + 8 * hi], n = s_in[f][1][a + 8 * hi];
        P = fma(c1[a], t, P);
        R = fma(c2[a], t, R);
        S = fma(c1[a], n, S);
      }
      s_tmp[f][0][lane] = P;
      s_tmp[f][1][lane] = R;
      s_tmp[f][2][lane] = S;
    }
    __syncthreads();
    // ---- pass 2: lane (a' = lo, b' = hi): contract the face index b
    {
      double c1[kFW], c2[kFW];
#pragma unroll
      for (int b = 0; b < kFW; ++b) { c1[b] = s_C[f][0][hi * 8 + b]; c2[b] = s_C[f][1][hi * 8 + b]; }
      double qu = 0.0, qtb = 0.0, qta = 0.0, qn = 0.0;
#pragma unroll
      for (int b = 0; b < kFW; ++b) {
        const double P = s_tmp[f][0][lo + 8 * b], R = s_tmp[f][1][lo + 8 * b], S = s_tmp[f][2][lo + 8 * b];
        qu = fma(c1[b], P, qu);
        qtb = fma(c2[b], P, qtb);
        qta = fma(c1[b], R, qta);
        qn = fma(c1[b], S, qn);
      }
      if (lo < NQ && hi < NQ) {
        double* out = qtrace + d.qoff + lo + NQ * hi;
        out[0] = qu;
        out[(1 + dir) * T] = qn;
        out[(1 + t0) * T] = qta;
        out[(1 + t1d) * T] = qtb;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// MFMA form of the trace kernel.  The two interpolation passes of a face are 8 x 8 x 8 matrix products; on the vector ALU
// every lane re-reads its operator row and its data column from LDS (45 KB of LDS traffic per wave and element: the kernel was
// LDS-bound, tools/pmc_faces.sh).  v_mfma_f64_16x16x4 takes the stacked operator [C; CD] (16 x 8) as a 2-register operand that
// never leaves the lane, so per element a lane only moves its own few values through LDS to re-shape them into MFMA operands:
//   pass 1:  Y (16 x 16) = [C; CD] (16 x 8) . [tr | nd] (8 x 16)         rows 0-7: C.tr | C.nd,  rows 8-15: CD.tr | (unused)
//   pass 2:  Z1 = [C.tr; CD.tr] (16 x 8) . [C^T | CD^T] (8 x 16)          -> qu | qtb ;  qta | (unused)
//            Z2 = [C.nd; 0]     (16 x 8) . [C^T | CD^T]                    -> qn
// Operand layouts (cdna_hip_programming.md): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D reg r: row (lane >> 4) + 4 r, col lane & 15.  [C^T | CD^T] as a B operand holds the same values as [C; CD] as an A operand.
// ---------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));


// one workgroup = 3 waves = the three reference directions of one element; wave `dir` serves the two faces 2 dir, 2 dir + 1,
// which share the same lines of u (trace = first / last entry, normal derivative = row 0 / row N-1 of D)
__global__ __launch_bounds__(192) void trace_mfma_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                         const SideDesc* __restrict__ sd, const ElemDesc* __restrict__ ed,
                                                         const double* __restrict__ face_ops, int n_elem,
                                                         const int* __restrict__ elist = nullptr) {
  // elist: the kernel works on the listed elements only (n_elem = length of the list): the elements with a ghost (+) side, whose traces
  // feed the exchange while the interior elements' operator kernel forms its traces from u itself (apply_operator)
  auto EL = [&](int i) { return elist ? elist[i] : i; };
  constexpr int LDM = 18;                 // padded row length of the 8 x 16 re-shaping buffer (<= 2-way bank conflicts)
  constexpr int UJ = 9, UK = 72;          // padded strides of the LDS copy of u: conflict-free face reads in all three directions
  constexpr int TPB = 192;
  __shared__ double s_u[8 * UK];
  __shared__ double s_x[3][2][8 * LDM];   // per wave and face: [a][(field, b)]
  // the direction is wave-uniform: with it in an SGPR the descriptor loads become scalar loads and cost no VGPRs
  const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lo = lane & 7, hi = lane >> 3;
  const int mi = lane & 15, mk = lane >> 4;   // MFMA operand coordinates of this lane
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
  int cur_off[2][2] = {{-1, -1}, {-1, -1}}, cur_offD = -1, cur_N = -1, cur_NQ[2] = {-1, -1};
  double opA[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, drow[2][kFW];
#pragma unroll
  for (int i = 0; i < kFW; ++i) drow[0][i] = drow[1][i] = 0.0;
  // persistent loop, software-pipelined: descriptors two elements ahead, u (3 values per thread) one element ahead
  int e = blockIdx.x;
  const int e_first = e < n_elem ? EL(e) : (n_elem > 0 ? EL(0) : 0);
  ElemDesc el = ed[e_first];
  SideDesc d0 = sd[6 * e_first + 2 * dir], d1 = sd[6 * e_first + 2 * dir + 1];
  double uv[3] = {0.0, 0.0, 0.0};
  {
    const int n3 = el.N * el.N * el.N;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (e < n_elem && (int)threadIdx.x + TPB * r < n3) uv[r] = u[el.ns + threadIdx.x + TPB * r];
  }
  ElemDesc edn = el;
  SideDesc dn0 = d0, dn1 = d1;
  if (e + (int)gridDim.x < n_elem) {
    const int e2 = EL(e + gridDim.x);
    edn = ed[e2];
    dn0 = sd[6 * e2 + 2 * dir];
    dn1 = sd[6 * e2 + 2 * dir + 1];
  }
  for (; e < n_elem; e += gridDim.x) {
    const int N = el.N, N2 = N * N, N3 = N2 * N;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int t = threadIdx.x + TPB * r;
      if (t < N3) s_u[(t % N) + UJ * ((t / N) % N) + UK * (t / N2)] = uv[r];
    }
    const ElemDesc el_next = edn;
    const SideDesc d0_next = dn0, d1_next = dn1;
    {
      const int en = e + gridDim.x;
      if (en < n_elem) {
        const int n3 = el_next.N * el_next.N * el_next.N;
#pragma unroll
        for (int r = 0; r < 3; ++r) uv[r] = ((int)threadIdx.x + TPB * r < n3) ? u[el_next.ns + threadIdx.x + TPB * r] : 0.0;
        const int en2 = en + gridDim.x;
        if (en2 < n_elem) {
          const int e2 = EL(en2);
          edn = ed[e2];
          dn0 = sd[6 * e2 + 2 * dir];
          dn1 = sd[6 * e2 + 2 * dir + 1];
        }
      }
    }
    // operator registers: reloaded only when the side's operators change (wave-uniform test)
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const SideDesc& d = s_ ? d1 : d0;
      if (d.offC != cur_off[s_][0] || d.offCD != cur_off[s_][1] || N != cur_N || d.NQ != cur_NQ[s_]) {
        const int row = mi & 7, off = (mi < 8) ? d.offC : d.offCD;
        opA[s_][0] = (row < d.NQ && mk < N) ? face_ops[off + row * N + mk] : 0.0;
        opA[s_][1] = (row < d.NQ && mk + 4 < N) ? face_ops[off + row * N + mk + 4] : 0.0;
        cur_off[s_][0] = d.offC; cur_off[s_][1] = d.offCD; cur_NQ[s_] = d.NQ;
      }
    }
    if (el.offD != cur_offD || N != cur_N) {
#pragma unroll
      for (int i = 0; i < kFW; ++i) {
        drow[0][i] = face_ops[el.offD + i];                 // zero-padded 8 x 8 image: row 0 and row N-1
        drow[1][i] = face_ops[el.offD + (N - 1) * 8 + i];
      }
      cur_offD = el.offD;
    }
    cur_N = N;
    __syncthreads();
    // ---- nodal traces and normal derivatives of both faces at lane (a = lo, b = hi): one pass over the line of u
    {
      const int sn = (dir == 0) ? 1 : (dir == 1 ? UJ : UK);
      const int sa = (dir == 0) ? UJ : 1, sb = (dir == 2) ? UJ : UK;
      double tr0 = 0.0, nd0 = 0.0, tr1 = 0.0, nd1 = 0.0;
      if (lo < N && hi < N) {
        const int base = lo * sa + hi * sb;
#pragma unroll
        for (int i = 0; i < kFW; ++i) {
          const double ui = s_u[base + (i < N ? i : N - 1) * sn];   // padded D columns are 0
          if (i == 0) tr0 = ui;
          if (i == N - 1) tr1 = ui;
          nd0 = fma(drow[0][i], ui, nd0);
          nd1 = fma(drow[1][i], ui, nd1);
        }
      }
      s_x[dir][0][lo * LDM + hi] = tr0;        // row k = a, column j = b
      s_x[dir][0][lo * LDM + 8 + hi] = nd0;    // column j = 8 + b
      s_x[dir][1][lo * LDM + hi] = tr1;
      s_x[dir][1][lo * LDM + 8 + hi] = nd1;
    }
    wave_lds_fence();
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const SideDesc& d = s_ ? d1 : d0;
      const int NQ = d.NQ, T = NQ * NQ;
      const double* sx = s_x[dir][s_];
      // ---- pass 1, transposed:  Y^T (16 x 16) = [tr | nd]^T (16 x 8) . [C^T | CD^T] (8 x 16)
      //      rows (field, b), columns (operator, a'); the operator B operand holds the same values as the A operand [C; CD]
      mfma_d4 y = {0.0, 0.0, 0.0, 0.0};
      {
        const double a0 = sx[mk * LDM + mi], a1 = sx[(mk + 4) * LDM + mi];   // A[i = (field, b)][k = a]
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, opA[s_][0], y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, opA[s_][1], y, 0, 0, 0);
      }
      // ---- pass 2:  Z = [C; CD] (16 x 8, along b) . Y^T rows.  The C/D registers of pass 1 ARE the B operands of pass 2:
      //      reg r holds row (lane >> 4) + 4 r = (field r >> 1, b = (lane >> 4) + 4 (r & 1)), i.e. k-step r & 1 of field r >> 1
      mfma_d4 z1 = {0.0, 0.0, 0.0, 0.0}, z2 = {0.0, 0.0, 0.0, 0.0};
      z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s_][0], y[0], z1, 0, 0, 0);
      z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s_][1], y[1], z1, 0, 0, 0);
      z2 = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s_][0], y[2], z2, 0, 0, 0);
      z2 = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s_][1], y[3], z2, 0, 0, 0);
      // ---- store: column lane & 15 = (operator along a, a'), reg r: row (lane >> 4) + 4 r = (operator along b, b')
      //      z1 rows C:  cols C -> qu, cols CD -> qta;  z1 rows CD: cols C -> qtb;  z2 rows C, cols C -> qn
      double* out = qtrace + d.qoff;
      const int aq = mi & 7;
      if (aq < NQ) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int bq = mk + 4 * r;
          if (bq < NQ) {
            out[((mi < 8) ? 0 : (1 + t0) * T) + aq + NQ * bq] = z1[r];
            if (mi < 8) {
              out[(1 + t1d) * T + aq + NQ * bq] = z1[2 + r];
              out[(1 + dir) * T + aq + NQ * bq] = z2[r];
            }
          }
        }
      }
    }
    __syncthreads();
    el = el_next; d0 = d0_next; d1 = d1_next;
  }
}

// The same kernel for 8 < N or NQ <= 16 (p = 8 .. 15): every 16 x 16 operand block is one MFMA tile, the contraction length
// N <= 16 is 4 k-steps.  Per face: pass 1 = 3 tiles (tr.C^T, tr.CD^T, nd.C^T) x 4 k-steps, pass 2 = 4 tiles (qu, qta, qtb, qn) x
// 4 k-steps = 28 MFMAs, all useful at N = 16.  Replaces the generic trace kernel, whose per-node LDS dot products made the
// face path five times more expensive than the volume kernel at p >= 9.
__global__ __launch_bounds__(192) void trace_mfma16_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                           const SideDesc* __restrict__ sd, const ElemDesc* __restrict__ ed,
                                                           const double* __restrict__ face_ops, int n_elem, int max_n,
                                                           const int* __restrict__ elist = nullptr) {
  constexpr int LDM = 34;                  // staging rows: 32 columns (field, b) + padding
  constexpr int UJ = 17, UK = 272;         // padded strides of the LDS copy of u (<= 2-way bank conflicts in all directions)
  constexpr int TPB = 192;
  extern __shared__ __attribute__((aligned(16))) double smem16[];
  double* s_u = smem16;                    // max_n * UK (only the k-planes the plan's largest element needs)
  const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* stage = smem16 + max_n * UK + dir * (2 * 16 * LDM);   // per wave: 2 faces x [a (16)][(field, b) 32]
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
  int cur_off[2][2] = {{-1, -1}, {-1, -1}}, cur_offD = -1, cur_N = -1, cur_NQ[2] = {-1, -1};
  double op[2][2][4];   // [face][C / CD][k-step]: OP[row mi][col 4 ks + mk], zero-padded
  double drow[2][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) drow[0][i] = drow[1][i] = 0.0;
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int c_ = 0; c_ < 4; ++c_) op[a_][b_][c_] = 0.0;
  // the staging rows / columns beyond N are never written again: zero them once (they meet zero operator entries, but LDS
  // garbage could be NaN)
  for (int i = lane; i < 2 * 16 * LDM; i += 64) stage[i] = 0.0;
  wave_lds_fence();
  for (int ei = blockIdx.x; ei < n_elem; ei += gridDim.x) {
    const int e = elist ? elist[ei] : ei;   // (elist: the listed elements only, see trace_mfma_kernel)
    const ElemDesc el = ed[e];
    const SideDesc d0 = sd[6 * e + 2 * dir], d1 = sd[6 * e + 2 * dir + 1];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    // (a rolled `for (t = tid; t < N3; t += TPB)` waits for each load in turn: N3 / TPB dependent memory round trips, 21 at p = 15)
    for (int t0 = threadIdx.x; t0 < N3; t0 += 8 * TPB) {
      double uv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) uv[c] = (t0 + c * TPB < N3) ? u[el.ns + t0 + c * TPB] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int t = t0 + c * TPB;
        if (t < N3) s_u[(t % N) + UJ * ((t / N) % N) + UK * (t / N2)] = uv[c];
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const SideDesc& d = s_ ? d1 : d0;
      if (d.offC != cur_off[s_][0] || d.offCD != cur_off[s_][1] || N != cur_N || d.NQ != cur_NQ[s_]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int col = 4 * ks + mk;
          const bool in = mi < d.NQ && col < N;
          op[s_][0][ks] = in ? face_ops[d.offC + mi * N + col] : 0.0;
          op[s_][1][ks] = in ? face_ops[d.offCD + mi * N + col] : 0.0;
        }
        cur_off[s_][0] = d.offC; cur_off[s_][1] = d.offCD; cur_NQ[s_] = d.NQ;
      }
    }
    if (el.offD != cur_offD || N != cur_N) {   // unpadded N x N derivative matrix: rows 0 and N-1
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        drow[0][i] = (i < N) ? face_ops[el.offD + i] : 0.0;
        drow[1][i] = (i < N) ? face_ops[el.offD + (N - 1) * N + i] : 0.0;
      }
      cur_offD = el.offD;
    }
    cur_N = N;
    __syncthreads();
    // ---- nodal traces / normal derivatives of both faces: 256 face nodes in 4 chunks of 64 lanes
    {
      const int sn = (dir == 0) ? 1 : (dir == 1 ? UJ : UK);
      const int sa = (dir == 0) ? UJ : 1, sb = (dir == 2) ? UJ : UK;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * c >= N) continue;   // wave-uniform: this chunk holds only padding
        const int idx = 64 * c + lane, a = idx & 15, b = idx >> 4;
        if (a < N && b < N) {
          double tr0 = 0.0, nd0 = 0.0, tr1 = 0.0, nd1 = 0.0;
          const int base = a * sa + b * sb;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i >= N) continue;   // wave-uniform
            const double ui = s_u[base + i * sn];
            if (i == 0) tr0 = ui;
            if (i == N - 1) tr1 = ui;
            nd0 = fma(drow[0][i], ui, nd0);
            nd1 = fma(drow[1][i], ui, nd1);
          }
          stage[a * LDM + b] = tr0;
          stage[a * LDM + 16 + b] = nd0;
          stage[16 * LDM + a * LDM + b] = tr1;
          stage[16 * LDM + a * LDM + 16 + b] = nd1;
        }
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const SideDesc& d = s_ ? d1 : d0;
      const int NQ = d.NQ, T = NQ * NQ;
      const double* st = stage + s_ * 16 * LDM;
      // pass 1 (transposed): rows (field, b), K = a, columns (operator, a')
      mfma_d4 ytc = {0.0, 0.0, 0.0, 0.0}, ytd = {0.0, 0.0, 0.0, 0.0}, ync = {0.0, 0.0, 0.0, 0.0};
      const int KN = (N + 3) >> 2;   // k-steps that hold data (wave-uniform): the others multiply zero padding
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= KN) continue;
        const double atr = st[(4 * ks + mk) * LDM + mi], and_ = st[(4 * ks + mk) * LDM + 16 + mi];
        ytc = __builtin_amdgcn_mfma_f64_16x16x4f64(atr, op[s_][0][ks], ytc, 0, 0, 0);
        ytd = __builtin_amdgcn_mfma_f64_16x16x4f64(atr, op[s_][1][ks], ytd, 0, 0, 0);
        ync = __builtin_amdgcn_mfma_f64_16x16x4f64(and_, op[s_][0][ks], ync, 0, 0, 0);
      }
      // pass 2: reg r of a pass-1 tile = rows 4 r .. 4 r + 3 (the b index) = B operand of k-step r
      mfma_d4 qu = {0.0, 0.0, 0.0, 0.0}, qta = {0.0, 0.0, 0.0, 0.0}, qtb = {0.0, 0.0, 0.0, 0.0}, qn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= KN) continue;
        qu = __builtin_amdgcn_mfma_f64_16x16x4f64(op[s_][0][r], ytc[r], qu, 0, 0, 0);
        qta = __builtin_amdgcn_mfma_f64_16x16x4f64(op[s_][0][r], ytd[r], qta, 0, 0, 0);
        qtb = __builtin_amdgcn_mfma_f64_16x16x4f64(op[s_][1][r], ytc[r], qtb, 0, 0, 0);
        qn = __builtin_amdgcn_mfma_f64_16x16x4f64(op[s_][0][r], ync[r], qn, 0, 0, 0);
      }
      // store: column mi = a', reg r: row mk + 4 r = b'
      double* out = qtrace + d.qoff;
      if (mi < NQ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bq = mk + 4 * r;
          if (bq < NQ) {
            out[mi + NQ * bq] = qu[r];
            out[(1 + t0) * T + mi + NQ * bq] = qta[r];
            out[(1 + t1d) * T + mi + NQ * bq] = qtb[r];
            out[(1 + dir) * T + mi + NQ * bq] = qn[r];
          }
        }
      }
    }
    __syncthreads();
  }
}

// FUSE: the Chebyshev update of the element (r = alpha (rhs - Au), p = r + beta p, u += p; cheby_update_kernel, same roundings) runs
// in the epilogue, where the element's final Au is in registers -- one kernel and one read of Au less per smoother iteration.  u is
// not an input of this kernel (the faces read traces), so updating it in place is safe.
// INPLACE: the per-wave operator apply works in place (23 KB of LDS per workgroup instead of 35 KB): faster on a mesh plan (config 2
// 41.6 -> 36.9 us, level 5 342 -> 316 us), slower on a Schwarz subdomain plan, where a third of the sides read one cached zero block and
// the kernel is bound by its LDS / issue chain rather than by memory latency (875 -> 962 us on 97 336 copies) -- chosen per plan.
template <bool FUSE, bool INPLACE>
__global__ __launch_bounds__(384, 6) void flux_wave_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                        double* __restrict__ Au, const SideDesc* __restrict__ sd,
                                                        const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                        const double* __restrict__ geom, const double* __restrict__ bndry_q,
                                                        const double* __restrict__ robin_c, const double* __restrict__ robin_r,
                                                        int n_elem, int xcd_chunk, ChebyFuse cf, const int* __restrict__ elist) {
  __shared__ double s_in[6][4][64];   // per wave: 4 term fields on the 8 x 8 grid
  __shared__ double s_tmp[INPLACE ? 1 : 6][INPLACE ? 1 : 4][64];
  __shared__ double s_E[6][64];       // per wave: E of its side, zero-padded to 8 x 8
  __shared__ double s_W[512];         // lifted face-local part: terms 1+3 and the tangential D^T of term 2
  __shared__ double s_N[6][64];       // per side: term 2 of the normal direction (D^T spreads it along the normal lines)
  __shared__ double s_D[64];          // zero-padded 8 x 8
  // the face index is wave-uniform: in an SGPR it turns the descriptor loads into scalar loads (no VGPRs for descriptors)
  const int f = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lo = lane & 7, hi = lane >> 3;
  // persistent workgroups: the descriptors of the NEXT element are requested while this one is computed
  // XCD-aware element order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the virtual index v is
  // mapped to element (v % 8) * chunk + v / 8: every XCD walks one contiguous (Morton-local) eighth of the elements and finds
  // its neighbours' traces in its own L2 more often.  xcd_chunk = 0: identity.
  // elist: the kernel works on the listed elements only (the hybrid operator's dirty elements; n_elem = the list's length)
  auto elem_of = [&](int v) { return elist ? elist[v] : (xcd_chunk > 0 ? (v & 7) * xcd_chunk + (v >> 3) : v); };
  int e = blockIdx.x;
  ElemDesc edn = ed[e < n_elem ? elem_of(e) : 0];
  SideDesc dn = sd[6 * (e < n_elem ? elem_of(e) : 0) + f];
  for (; e < n_elem; e += gridDim.x) {
    const ElemDesc el = edn;
    const SideDesc d = dn;
    {
      const int en = e + gridDim.x;
      if (en < n_elem) {
        edn = ed[elem_of(en)];
        dn = sd[6 * elem_of(en) + f];
      }
    }
    const int N = el.N, N2 = N * N, N3 = N2 * N, NQ = d.NQ, T = NQ * NQ;
    const bool on_m = lo < N && hi < N, on_q = lo < NQ && hi < NQ;
    // ---- every global load of this side up front (independent requests: one memory latency)
    double qm[4] = {0, 0, 0, 0}, qp[4] = {0, 0, 0, 0}, gq[7] = {0, 0, 0, 0, 0, 0, 0};
    if (on_q && d.kind != 3) {   // kind 3: a hanging side of an hp-split plan -- its terms come from the mortar-record kernel (all-zero inputs here)
      const int k = lo + NQ * hi;
      const double* m = qtrace + d.qoff + k;
#pragma unroll
      for (int c = 0; c < 4; ++c) qm[c] = m[c * T];
      if (d.kind != 0) {
        const double* p = ((d.kind == 2) ? ghost_qtrace : qtrace) + d.nbr_qoff + reorder_index(d.code, NQ - 1, lo, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) qp[c] = p[c * T];
      } else if (robin_c) {
        qp[0] = robin_r[d.geom + k];
      } else {
        qp[0] = bndry_q[d.geom + k];
      }
      if (d.kind == 0 && robin_c) {
        gq[6] = robin_c[d.geom + k];  // am = ap = 0: no term 1 / term 2 on a Robin side
      } else {
        const double* g = geom + (size_t)7 * d.geom + k;
#pragma unroll
        for (int c = 0; c < 7; ++c) gq[c] = g[c * T];
      }
    }
    const double ope = (hi < N && lo < NQ) ? face_ops[d.offE + hi * NQ + lo] : 0.0;
    const double dval = (threadIdx.x < 64) ? face_ops[el.offD + threadIdx.x] : 0.0;
    // Au_e is read now (2 values per thread) so that the final update only stores
    double au0 = 0.0, au1 = 0.0;
    if ((int)threadIdx.x < N3) au0 = Au[el.ns + threadIdx.x];
    if ((int)threadIdx.x + 384 < N3) au1 = Au[el.ns + threadIdx.x + 384];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) s_W[i] = 0.0;
    s_E[f][lane] = ope;
    if (threadIdx.x < 64) s_D[threadIdx.x] = dval;
    // ---- SIPG terms at the quadrature node of this lane (padding lanes carry zeros)
    {
      double t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) t1 += gq[i] * qm[1 + i] + gq[3 + i] * qp[1 + i];  // gq[3..5] = 0 on boundary sides
      const double jump = qm[0] - qp[0];
      const double w1 = (d.kind != 0) ? -0.5 : -1.0;
      // Robin boundary (d4est_laplacian_flux_sipg.c:339-489): sj (coeff u_m - rhs) only
      s_in[f][0][lane] = (d.kind == 0 && robin_c) ? gq[6] * qm[0] - qp[0] : w1 * t1 + gq[6] * jump;
#pragma unroll
      for (int l = 0; l < 3; ++l) s_in[f][1 + l][lane] = w1 * gq[l] * jump;
    }
    __syncthreads();
    // ---- integrate and project onto the (-) side: 4 fields on the N x N face nodes
    double res[4];
    if (INPLACE) wave_apply2d<4>(s_E[f], &s_in[f][0][0], lane, res);
    else wave_apply2d_tmp<4>(s_E[f], &s_in[f][0][0], &s_tmp[INPLACE ? 0 : f][0][0], lane, res);
    // ---- D^T of the two TANGENTIAL term-2 fields stays inside the face: val = t13 + D_a^T t2_a + D_b^T t2_b
    const int dir = f >> 1;
    const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;  // reference directions of the face indices a and b
    s_in[f][0][lane] = res[1 + t0];
    s_in[f][1][lane] = res[1 + t1d];
    s_N[f][lane] = res[1 + dir];
    __syncthreads();
    double val = res[0];
#pragma unroll
    for (int q = 0; q < kFW; ++q) {
      val = fma(s_D[q * 8 + lo], s_in[f][0][q + 8 * hi], val);   // padded rows/columns of D are zero
      val = fma(s_D[q * 8 + hi], s_in[f][1][lo + 8 * q], val);
    }
    // ---- lift the face-local part, opposite faces together (disjoint node sets)
    for (int phase = 0; phase < 3; ++phase) {
      if (dir == phase && on_m) s_W[face_vol_index(f, N, lo, hi)] += val;
      __syncthreads();
    }
    // ---- Au_e += W + sum over the six sides of D[fix][.] (x) N_f  (the normal D^T touches the whole line)
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int idx = threadIdx.x + rep * 384;
      if (idx < N3) {
        const int i = idx % N, j = (idx / N) % N, k = idx / N2;
        double v = s_W[idx];
        v = fma(s_D[i], s_N[0][j + 8 * k], v);
        v = fma(s_D[(N - 1) * 8 + i], s_N[1][j + 8 * k], v);
        v = fma(s_D[j], s_N[2][i + 8 * k], v);
        v = fma(s_D[(N - 1) * 8 + j], s_N[3][i + 8 * k], v);
        v = fma(s_D[k], s_N[4][i + 8 * j], v);
        v = fma(s_D[(N - 1) * 8 + k], s_N[5][i + 8 * j], v);
        const double a = (rep == 0 ? au0 : au1) + v;
        Au[el.ns + idx] = a;
        if (FUSE) {
          const size_t o = (size_t)el.ns + idx;
          const double res = __dadd_rn(cf.rhs[o], __dmul_rn(-1.0, a));
          const double ri = __dmul_rn(cf.alpha, res);
          const double pi = __dadd_rn(__dmul_rn(cf.beta, cf.p[o]), ri);
          if (cf.r) cf.r[o] = ri;
          cf.p[o] = pi;
          (cf.u_out ? cf.u_out : cf.u)[o] = __dadd_rn(cf.u[o], pi);   // (u_out: the hybrid operator's second vector -- its clean kernels read the neighbours' u)
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Tiled MFMA flux kernel for N, NQ <= 16 (the p = 8 .. 15 counterpart of flux_wave_kernel).  One workgroup = 3 waves = the three
// reference directions of one element; wave `dir` serves faces 2 dir and 2 dir + 1.  Per face:
//   terms   lane (mi, mk) evaluates the 4 SIPG term fields at its 4 mortar nodes (a' = mi, b' = 4 ks + mk) -- exactly the values
//           it needs as the A operand of the first contraction, so the term fields never touch LDS
//   pass 1  Y_c = A_c . E^T   (contract b', 4 k-steps, tile: column = side index b, rows = a')
//   pass 2  R_c = E . Y_c     (contract a'; the tile registers of pass 1 are the B operands)            32 MFMAs for 4 fields
//   D^T     val = R_0 + D^T . R_t0  (tile registers again)  +  R_t1 . D  (one tile transposed through LDS)    8 MFMAs
// then the face-local part `val` and the normal term-2 field (whose D^T spreads along the whole normal line) of the six faces
// are combined into Au_e by all threads from LDS tiles.
// ---------------------------------------------------------------------------
template <bool FUSE>   // FUSE: Chebyshev update in the epilogue, see flux_wave_kernel
__global__ __launch_bounds__(192) void flux_mfma16_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                          double* __restrict__ Au, const SideDesc* __restrict__ sd,
                                                          const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                          const double* __restrict__ geom, const double* __restrict__ bndry_q,
                                                          const double* __restrict__ robin_c, const double* __restrict__ robin_r,
                                                          int n_elem, int xcd_chunk, ChebyFuse cf, const int* __restrict__ elist) {
  constexpr int LT = 17;                    // padded row length of a 16 x 16 tile in LDS
  constexpr int TPB = 192;
  __shared__ double s_tile[6][2][16 * LT];  // per face: val, normal field   (rows a, columns b)
  __shared__ double s_tr[3][16 * LT];       // per wave: transposition buffer
  __shared__ double s_Dfix[6][16];          // per face: row `fix` of D (the normal D^T)
  const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;   // reference directions of the face indices a and b
  int cur_offE[2] = {-1, -1}, cur_offD = -1, cur_N = -1, cur_NQ[2] = {-1, -1};
  double opE[2][4], opD[4];   // E[mi][4 ks + mk] (N x NQ) per face;  D[4 ks + mk][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i) opE[0][i] = opE[1][i] = opD[i] = 0.0;
  for (int v = blockIdx.x; v < n_elem; v += gridDim.x) {
    const int e = elist ? elist[v] : (xcd_chunk > 0 ? (v & 7) * xcd_chunk + (v >> 3) : v);   // XCD-aware element order / element list, see flux_wave_kernel
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    if (el.offD != cur_offD || N != cur_N) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) opD[ks] = (4 * ks + mk < N && mi < N) ? face_ops[el.offD + (4 * ks + mk) * N + mi] : 0.0;
      cur_offD = el.offD;
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int f = 2 * dir + s_;
      const SideDesc d = sd[6 * e + f];
      const int NQ = d.NQ, T = NQ * NQ;
      if (d.offE != cur_offE[s_] || N != cur_N || NQ != cur_NQ[s_]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) opE[s_][ks] = (mi < N && 4 * ks + mk < NQ) ? face_ops[d.offE + mi * NQ + 4 * ks + mk] : 0.0;
        cur_offE[s_] = d.offE; cur_NQ[s_] = NQ;
      }
      if (lane < 16) s_Dfix[f][lane] = (lane < N) ? face_ops[el.offD + face_fix(f, N) * N + lane] : 0.0;
      // ---- SIPG terms at this lane's 4 mortar nodes (a' = mi, b' = 4 ks + mk): the A operands of pass 1
      double A[4][4];
      const bool robin = (d.kind == 0) && robin_c;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int bq = 4 * ks + mk;
        A[0][ks] = A[1][ks] = A[2][ks] = A[3][ks] = 0.0;
        if (mi < NQ && bq < NQ && d.kind != 3) {   // kind 3: a hanging side of an hp-split plan -- its terms come from the mortar-record kernels
          const int k = mi + NQ * bq;
          const double* m = qtrace + d.qoff + k;
          const double um = m[0];
          if (robin) {
            A[0][ks] = robin_c[d.geom + k] * um - robin_r[d.geom + k];
          } else {
            const double* g = geom + (size_t)7 * d.geom + k;
            double up, t1 = 0.0, am[3];
            if (d.kind != 0) {
              const double* pp = ((d.kind == 2) ? ghost_qtrace : qtrace) + d.nbr_qoff + reorder_index(d.code, NQ - 1, mi, bq);
              up = pp[0];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                am[i] = g[i * T];
                t1 += am[i] * m[(1 + i) * T] + g[(3 + i) * T] * pp[(1 + i) * T];
              }
            } else {
              up = bndry_q[d.geom + k];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                am[i] = g[i * T];
                t1 += am[i] * m[(1 + i) * T];
              }
            }
            const double jump = um - up;
            const double w1 = (d.kind != 0) ? -0.5 : -1.0;
            A[0][ks] = w1 * t1 + g[6 * T] * jump;
            A[1][ks] = w1 * am[t0] * jump;    // field order: tangential a, tangential b, normal
            A[2][ks] = w1 * am[t1d] * jump;
            A[3][ks] = w1 * am[dir] * jump;
          }
        }
      }
      // ---- pass 1 / pass 2 per field
      mfma_d4 R[4];
      const int KQ = (NQ + 3) >> 2, KN = (N + 3) >> 2;   // k-steps that hold data (wave-uniform)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mfma_d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < KQ) y = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c][ks], opE[s_][ks], y, 0, 0, 0);
        mfma_d4 rr = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < KQ) rr = __builtin_amdgcn_mfma_f64_16x16x4f64(opE[s_][r], y[r], rr, 0, 0, 0);
        R[c] = rr;   // tile: column mi = b, reg r: row mk + 4 r = a
      }
      // ---- val = R_0 + D_a^T R_ta + R_tb D
      mfma_d4 val = R[0];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < KN) val = __builtin_amdgcn_mfma_f64_16x16x4f64(opD[r], R[1][r], val, 0, 0, 0);   // A = D^T: D[4 r + mk][mi]
      {
        double* tb = s_tr[dir];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(mk + 4 * r) * LT + mi] = R[2][r];
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks >= KN) continue;
          const double a_ = tb[mi * LT + 4 * ks + mk];            // A[i = a][k = q]
          val = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, opD[ks], val, 0, 0, 0);   // B = D[k = q][j = b]
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_tile[f][0][(mk + 4 * r) * LT + mi] = val[r];
        s_tile[f][1][(mk + 4 * r) * LT + mi] = R[3][r];
      }
    }
    cur_N = N;
    __syncthreads();
    // ---- Au_e += lift(val_f) + D[fix_f][.] (x) Nrm_f over the six faces
    // (Au and the smoother vectors of six nodes per thread are requested before the first is used: the rolled loop waited for its
    // loads every trip)
    for (int idx0 = threadIdx.x; idx0 < N3; idx0 += 6 * TPB) {
      double au_[6], rh_[6], pp_[6], uu_[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int idx = idx0 + c * TPB;
        au_[c] = rh_[c] = pp_[c] = uu_[c] = 0.0;
        if (idx < N3) {
          const size_t o = (size_t)el.ns + idx;
          au_[c] = Au[o];
          if (FUSE) {
            rh_[c] = cf.rhs[o];
            pp_[c] = cf.p[o];
            uu_[c] = cf.u[o];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
      const int idx = idx0 + c * TPB;
      if (idx >= N3) continue;
      const int i = idx % N, j = (idx / N) % N, k = idx / N2;
      double v = 0.0;
      v = fma(s_Dfix[0][i], s_tile[0][1][j * LT + k], v);
      v = fma(s_Dfix[1][i], s_tile[1][1][j * LT + k], v);
      v = fma(s_Dfix[2][j], s_tile[2][1][i * LT + k], v);
      v = fma(s_Dfix[3][j], s_tile[3][1][i * LT + k], v);
      v = fma(s_Dfix[4][k], s_tile[4][1][i * LT + j], v);
      v = fma(s_Dfix[5][k], s_tile[5][1][i * LT + j], v);
      if (i == 0) v += s_tile[0][0][j * LT + k];
      if (i == N - 1) v += s_tile[1][0][j * LT + k];
      if (j == 0) v += s_tile[2][0][i * LT + k];
      if (j == N - 1) v += s_tile[3][0][i * LT + k];
      if (k == 0) v += s_tile[4][0][i * LT + j];
      if (k == N - 1) v += s_tile[5][0][i * LT + j];
      const size_t o = (size_t)el.ns + idx;
      const double a = au_[c] + v;
      Au[o] = a;
      if (FUSE) {
        const double res = __dadd_rn(rh_[c], __dmul_rn(-1.0, a));
        const double ri = __dmul_rn(cf.alpha, res);
        const double pi = __dadd_rn(__dmul_rn(cf.beta, pp_[c]), ri);
        if (cf.r) cf.r[o] = ri;
        cf.p[o] = pi;
        (cf.u_out ? cf.u_out : cf.u)[o] = __dadd_rn(uu_[c], pi);
      }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Mortar-record (hanging-face) counterparts of the tiled MFMA kernels, N, NQ <= 16: separate 1-D operators along the two
// face axes (the hp-prolongation child of a big side differs per axis), 1 or 4 records per side; the nodal trace of a big
// side is formed once and interpolated onto its four sub-mortars, and the four sub-mortar contributions of a big side are
// summed in the MFMA accumulators before the lift.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(192) void trace_hp_mfma16_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                              const HpMortar* __restrict__ md, const int* __restrict__ side_first,
                                                              const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                              const double* __restrict__ hp_ops, int n_elem, int max_n,
                                                              const int* __restrict__ elist = nullptr, int hang_only = 0) {
  // elist / hang_only (hp split, launch_traces): the kernel walks the listed elements -- those with a hanging side -- and serves the
  // records of their HANGING sides only; the conforming sides of the whole mesh are the fast conforming kernel's
  constexpr int LDM = 34;
  constexpr int UJ = 17, UK = 272;
  constexpr int TPB = 192;
  extern __shared__ __attribute__((aligned(16))) double smem16[];
  double* s_u = smem16;
  const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* stage = smem16 + max_n * UK + dir * (2 * 16 * LDM);
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
  int cur_offD = -1, cur_N = -1;
  double drow[2][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) drow[0][i] = drow[1][i] = 0.0;
  for (int i = lane; i < 2 * 16 * LDM; i += 64) stage[i] = 0.0;
  wave_lds_fence();
  for (int ei = blockIdx.x; ei < n_elem; ei += gridDim.x) {
    const int e = elist ? elist[ei] : ei;
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    // (a rolled `for (t = tid; t < N3; t += TPB)` waits for each load in turn: N3 / TPB dependent memory round trips, 21 at p = 15)
    for (int t0 = threadIdx.x; t0 < N3; t0 += 8 * TPB) {
      double uv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) uv[c] = (t0 + c * TPB < N3) ? u[el.ns + t0 + c * TPB] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int t = t0 + c * TPB;
        if (t < N3) s_u[(t % N) + UJ * ((t / N) % N) + UK * (t / N2)] = uv[c];
      }
    }
    if (el.offD != cur_offD || N != cur_N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        drow[0][i] = (i < N) ? face_ops[el.offD + i] : 0.0;
        drow[1][i] = (i < N) ? face_ops[el.offD + (N - 1) * N + i] : 0.0;
      }
      cur_offD = el.offD;
      cur_N = N;
    }
    __syncthreads();
    {
      const int sn = (dir == 0) ? 1 : (dir == 1 ? UJ : UK);
      const int sa = (dir == 0) ? UJ : 1, sb = (dir == 2) ? UJ : UK;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * c >= N) continue;
        const int idx = 64 * c + lane, a = idx & 15, b = idx >> 4;
        if (a < N && b < N) {
          double tr0 = 0.0, nd0 = 0.0, tr1 = 0.0, nd1 = 0.0;
          const int base = a * sa + b * sb;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i >= N) continue;
            const double ui = s_u[base + i * sn];
            if (i == 0) tr0 = ui;
            if (i == N - 1) tr1 = ui;
            nd0 = fma(drow[0][i], ui, nd0);
            nd1 = fma(drow[1][i], ui, nd1);
          }
          stage[a * LDM + b] = tr0;
          stage[a * LDM + 16 + b] = nd0;
          stage[16 * LDM + a * LDM + b] = tr1;
          stage[16 * LDM + a * LDM + 16 + b] = nd1;
        }
      }
    }
    wave_lds_fence();
    const int KN = (N + 3) >> 2;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int sidx = 6 * e + 2 * dir + s_;
      const double* st = stage + s_ * 16 * LDM;
      double aval[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        aval[0][ks] = st[(4 * ks + mk) * LDM + mi];
        aval[1][ks] = st[(4 * ks + mk) * LDM + 16 + mi];
      }
      for (int r_ = side_first[sidx]; r_ < side_first[sidx + 1]; ++r_) {
        const HpMortar m = md[r_];
        if (hang_only && m.hang == 0.0) continue;   // (wave-uniform)
        const int NQ = m.NQ, T = NQ * NQ;
        double oa[2][4], ob[2][4];   // along a: C, CD ; along b: C, CD   -- OP[row mi][col 4 ks + mk]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int col = 4 * ks + mk;
          const bool in = mi < NQ && col < N;
          oa[0][ks] = in ? hp_ops[m.offCa + mi * N + col] : 0.0;
          oa[1][ks] = in ? hp_ops[m.offCDa + mi * N + col] : 0.0;
          ob[0][ks] = in ? hp_ops[m.offCb + mi * N + col] : 0.0;
          ob[1][ks] = in ? hp_ops[m.offCDb + mi * N + col] : 0.0;
        }
        mfma_d4 ytc = {0.0, 0.0, 0.0, 0.0}, ytd = {0.0, 0.0, 0.0, 0.0}, ync = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks >= KN) continue;
          ytc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[0][ks], oa[0][ks], ytc, 0, 0, 0);
          ytd = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[0][ks], oa[1][ks], ytd, 0, 0, 0);
          ync = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[1][ks], oa[0][ks], ync, 0, 0, 0);
        }
        mfma_d4 qu = {0.0, 0.0, 0.0, 0.0}, qta = {0.0, 0.0, 0.0, 0.0}, qtb = {0.0, 0.0, 0.0, 0.0}, qn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r >= KN) continue;
          qu = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ytc[r], qu, 0, 0, 0);
          qta = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ytd[r], qta, 0, 0, 0);
          qtb = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[1][r], ytc[r], qtb, 0, 0, 0);
          qn = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ync[r], qn, 0, 0, 0);
        }
        double* out = qtrace + m.qoff;
        if (mi < NQ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int bq = mk + 4 * r;
            if (bq < NQ) {
              out[mi + NQ * bq] = qu[r];
              out[(1 + t0) * T + mi + NQ * bq] = qta[r];
              out[(1 + t1d) * T + mi + NQ * bq] = qtb[r];
              out[(1 + dir) * T + mi + NQ * bq] = qn[r];
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(192) void flux_hp_mfma16_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                             double* __restrict__ Au,
                                                             const HpMortar* __restrict__ md, const int* __restrict__ side_first,
                                                             const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                             const double* __restrict__ hp_ops, const double* __restrict__ geom,
                                                             const double* __restrict__ bndry_q, const double* __restrict__ robin_c,
                                                             const double* __restrict__ robin_r, int n_elem,
                                                             const int* __restrict__ elist = nullptr, int hang_only = 0) {
  // elist / hang_only: see trace_hp_mfma16_kernel (the conforming sides' terms were added by the fast conforming flux kernel)
  constexpr int LT = 17;
  constexpr int TPB = 192;
  __shared__ double s_tile[6][2][16 * LT];
  __shared__ double s_tr[3][16 * LT];
  __shared__ double s_Dfix[6][16];
  const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
  int cur_offD = -1, cur_N = -1;
  double opD[4] = {0.0, 0.0, 0.0, 0.0};
  for (int ei = blockIdx.x; ei < n_elem; ei += gridDim.x) {
    const int e = elist ? elist[ei] : ei;
    const ElemDesc el = ed[e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    if (el.offD != cur_offD || N != cur_N) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) opD[ks] = (4 * ks + mk < N && mi < N) ? face_ops[el.offD + (4 * ks + mk) * N + mi] : 0.0;
      cur_offD = el.offD;
      cur_N = N;
    }
    const int KN = (N + 3) >> 2;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int f = 2 * dir + s_, sidx = 6 * e + f;
      if (lane < 16) s_Dfix[f][lane] = (lane < N) ? face_ops[el.offD + face_fix(f, N) * N + lane] : 0.0;
      mfma_d4 R[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c] = mfma_d4{0.0, 0.0, 0.0, 0.0};
      for (int r_ = side_first[sidx]; r_ < side_first[sidx + 1]; ++r_) {
        const HpMortar m = md[r_];
        if (hang_only && m.hang == 0.0) continue;   // (wave-uniform)
        const int NQ = m.NQ, T = NQ * NQ, KQ = (NQ + 3) >> 2;
        double oea[4], oeb[4];   // E along a / b: E[mi][4 ks + mk]  (N x NQ)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bool in = mi < N && 4 * ks + mk < NQ;
          oea[ks] = in ? hp_ops[m.offEa + mi * NQ + 4 * ks + mk] : 0.0;
          oeb[ks] = in ? hp_ops[m.offEb + mi * NQ + 4 * ks + mk] : 0.0;
        }
        double A[4][4];
        const bool robin = (m.kind == 0) && robin_c;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int bq = 4 * ks + mk;
          A[0][ks] = A[1][ks] = A[2][ks] = A[3][ks] = 0.0;
          if (mi < NQ && bq < NQ) {
            const int k = mi + NQ * bq;
            const double* qm = qtrace + m.qoff + k;
            const double um = qm[0];
            if (robin) {
              A[0][ks] = robin_c[m.gidx + k] * um - robin_r[m.gidx + k];
            } else {
              const double* g = geom + (size_t)7 * m.gidx + k;
              double up, tm = 0.0, tp = 0.0, am[3];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                am[i] = g[i * T];
                tm += am[i] * qm[(1 + i) * T];
              }
              if (m.kind != 0) {
                const double* pp = ((m.kind == 2) ? ghost_qtrace : qtrace) + m.nbr_qoff + reorder_index(m.code, NQ - 1, mi, bq);
                up = pp[m.u_shift];
#pragma unroll
                for (int i = 0; i < 3; ++i) tp += g[(3 + i) * T] * pp[(1 + i) * T];
              } else {
                up = bndry_q[m.gidx + k];
              }
              const double jump = um - up;
              const double w1 = (m.kind != 0) ? -0.5 : -1.0;
              A[0][ks] = w1 * (tm + tp) + g[6 * T] * jump;   // (fm, fp, w2: folded into the factors by face_geom_hp_kernel)
              A[1][ks] = w1 * am[t0] * jump;
              A[2][ks] = w1 * am[t1d] * jump;
              A[3][ks] = w1 * am[dir] * jump;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          mfma_d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (ks < KQ) y = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c][ks], oeb[ks], y, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < KQ) R[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(oea[r], y[r], R[c], 0, 0, 0);   // summed over the side's mortars
        }
      }
      mfma_d4 val = R[0];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < KN) val = __builtin_amdgcn_mfma_f64_16x16x4f64(opD[r], R[1][r], val, 0, 0, 0);
      {
        double* tb = s_tr[dir];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(mk + 4 * r) * LT + mi] = R[2][r];
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks >= KN) continue;
          const double a_ = tb[mi * LT + 4 * ks + mk];
          val = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, opD[ks], val, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_tile[f][0][(mk + 4 * r) * LT + mi] = val[r];
        s_tile[f][1][(mk + 4 * r) * LT + mi] = R[3][r];
      }
    }
    __syncthreads();
    for (int idx0 = threadIdx.x; idx0 < N3; idx0 += 8 * TPB) {
      double au_[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) au_[c] = (idx0 + c * TPB < N3) ? Au[el.ns + idx0 + c * TPB] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
      const int idx = idx0 + c * TPB;
      if (idx >= N3) continue;
      const int i = idx % N, j = (idx / N) % N, k = idx / N2;
      double v = 0.0;
      v = fma(s_Dfix[0][i], s_tile[0][1][j * LT + k], v);
      v = fma(s_Dfix[1][i], s_tile[1][1][j * LT + k], v);
      v = fma(s_Dfix[2][j], s_tile[2][1][i * LT + k], v);
      v = fma(s_Dfix[3][j], s_tile[3][1][i * LT + k], v);
      v = fma(s_Dfix[4][k], s_tile[4][1][i * LT + j], v);
      v = fma(s_Dfix[5][k], s_tile[5][1][i * LT + j], v);
      if (i == 0) v += s_tile[0][0][j * LT + k];
      if (i == N - 1) v += s_tile[1][0][j * LT + k];
      if (j == 0) v += s_tile[2][0][i * LT + k];
      if (j == N - 1) v += s_tile[3][0][i * LT + k];
      if (k == 0) v += s_tile[4][0][i * LT + j];
      if (k == N - 1) v += s_tile[5][0][i * LT + j];
      Au[el.ns + idx] = au_[c] + v;
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------
// hp split, the record kernels' work in its parallel form: a UNIT is one side that stays with the mortar records (a big hanging side: 4
// records; a small side the conforming kernels cannot take: 1), and its records go to the 4 wavefronts of a workgroup side by side
// instead of one after the other.  The two kernels above walk an element's sides in three direction-waves and a side's records in
// a loop: on a locally refined mesh (one hanging side per listed element, a few hundred listed elements) two of the three waves find
// nothing to do and the third runs four dependent chains of descriptor load -> operator / trace loads -> MFMA one after another
// (13.8 + 22.7 us per apply at level 4, p = 7, 350 listed elements: pure latency).  Same arithmetic per record; the four sub-mortar
// contributions of a big side meet in LDS and are summed in record order (deterministic).
// ---------------------------------------------------------------------------
struct HangUnit { int e, f, r0, nrec; };

__global__ __launch_bounds__(256) void trace_unit_kernel(const double* __restrict__ u, double* __restrict__ qtrace,
                                                         const HpMortar* __restrict__ md, const HangUnit* __restrict__ units,
                                                         const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                         const double* __restrict__ hp_ops, int n_units, int max_n) {
  constexpr int LDM = 34;
  constexpr int TPB = 256;
  // the element's u as [i + UJ (j + ...)] with odd row and slab strides sized by the plan's largest degree (a p = 7 plan: 5.3 KB instead of
  // the 17.4 KB of the 16-wide image: workgroups per CU are what bounds this kernel on densely refined meshes)
  const int UJ = max_n | 1, UK = (max_n * UJ) | 1;
  extern __shared__ __attribute__((aligned(16))) double smem16[];
  double* s_u = smem16;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* stage = smem16 + ((max_n * UK + 1) & ~1) + wv * (16 * LDM);   // per wave: the side's nodal trace (columns 0..15) and normal derivative (16..31)
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  for (int i = lane; i < 16 * LDM; i += 64) stage[i] = 0.0;
  for (int ui = blockIdx.x; ui < n_units; ui += gridDim.x) {
    const HangUnit un = units[ui];
    const ElemDesc el = ed[un.e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    const int dir = un.f >> 1, hi = un.f & 1;
    const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
    // this wave's record, requested with the element's data (its operator offsets are needed two round trips later)
    const bool have = wv < un.nrec;
    const HpMortar m = md[un.r0 + (have ? wv : 0)];
    for (int tb = threadIdx.x; tb < N3; tb += 8 * TPB) {
      double uv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) uv[c] = (tb + c * TPB < N3) ? u[el.ns + tb + c * TPB] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int t = tb + c * TPB;
        if (t < N3) s_u[(t % N) + UJ * ((t / N) % N) + UK * (t / N2)] = uv[c];
      }
    }
    double drow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) drow[i] = (i < N) ? face_ops[el.offD + (hi ? (N - 1) * N : 0) + i] : 0.0;
    const int NQ = m.NQ, T = NQ * NQ;
    double oa[2][4], ob[2][4];   // along a: C, CD ; along b: C, CD   -- OP[row mi][col 4 ks + mk]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int col = 4 * ks + mk;
      const bool in = have && mi < NQ && col < N;
      oa[0][ks] = in ? hp_ops[m.offCa + mi * N + col] : 0.0;
      oa[1][ks] = in ? hp_ops[m.offCDa + mi * N + col] : 0.0;
      ob[0][ks] = in ? hp_ops[m.offCb + mi * N + col] : 0.0;
      ob[1][ks] = in ? hp_ops[m.offCDb + mi * N + col] : 0.0;
    }
    __syncthreads();
    if (have) {
      const int sn = (dir == 0) ? 1 : (dir == 1 ? UJ : UK);
      const int sa = (dir == 0) ? UJ : 1, sb = (dir == 2) ? UJ : UK;
      const int fixo = hi ? (N - 1) * sn : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * c >= N) continue;
        const int idx = 64 * c + lane, a = idx & 15, b = idx >> 4;
        if (a < N && b < N) {
          double nd = 0.0;
          const int base = a * sa + b * sb;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i >= N) continue;
            nd = fma(drow[i], s_u[base + i * sn], nd);
          }
          stage[a * LDM + b] = s_u[base + fixo];
          stage[a * LDM + 16 + b] = nd;
        }
      }
      wave_lds_fence();
      const int KN = (N + 3) >> 2;
      double aval[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        aval[0][ks] = stage[(4 * ks + mk) * LDM + mi];
        aval[1][ks] = stage[(4 * ks + mk) * LDM + 16 + mi];
      }
      mfma_d4 ytc = {0.0, 0.0, 0.0, 0.0}, ytd = {0.0, 0.0, 0.0, 0.0}, ync = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= KN) continue;
        ytc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[0][ks], oa[0][ks], ytc, 0, 0, 0);
        ytd = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[0][ks], oa[1][ks], ytd, 0, 0, 0);
        ync = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[1][ks], oa[0][ks], ync, 0, 0, 0);
      }
      mfma_d4 qu = {0.0, 0.0, 0.0, 0.0}, qta = {0.0, 0.0, 0.0, 0.0}, qtb = {0.0, 0.0, 0.0, 0.0}, qn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= KN) continue;
        qu = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ytc[r], qu, 0, 0, 0);
        qta = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ytd[r], qta, 0, 0, 0);
        qtb = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[1][r], ytc[r], qtb, 0, 0, 0);
        qn = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[0][r], ync[r], qn, 0, 0, 0);
      }
      double* out = qtrace + m.qoff;
      if (mi < NQ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bq = mk + 4 * r;
          if (bq < NQ) {
            out[mi + NQ * bq] = qu[r];
            out[(1 + t0) * T + mi + NQ * bq] = qta[r];
            out[(1 + t1d) * T + mi + NQ * bq] = qtb[r];
            out[(1 + dir) * T + mi + NQ * bq] = qn[r];
          }
        }
      }
    }
    __syncthreads();
  }
}

// one workgroup per listed element; its units one after the other, a unit's records on the 4 wavefronts.
// FUSE: the Chebyshev update of the listed elements in the epilogue (their A u is final only here: hybrid operator, hanging-aware form;
// roundings of cheby_update_kernel; the new iterate goes to cf.u_out -- the operator kernel reads the neighbours' u)
// TS: tile size of the LDS images (8 where every degree of the plan is <= 7: 15 KB instead of 55 KB per workgroup -- on densely refined
// meshes the workgroups per CU bound this kernel --, else 16)
template <bool FUSE, int TS>
__global__ __launch_bounds__(256) void flux_unit_kernel(const double* __restrict__ qtrace, const double* __restrict__ ghost_qtrace,
                                                        double* __restrict__ Au, const HpMortar* __restrict__ md,
                                                        const HangUnit* __restrict__ units, const int* __restrict__ unit_first,
                                                        const ElemDesc* __restrict__ ed, const double* __restrict__ face_ops,
                                                        const double* __restrict__ hp_ops, const double* __restrict__ geom,
                                                        const double* __restrict__ bndry_q, const double* __restrict__ robin_c,
                                                        const double* __restrict__ robin_r, int n_elem, ChebyFuse cf) {
  constexpr int LT = TS + 1;
  constexpr int TPB = 256;
  constexpr int NAU = (TS <= 8) ? 2 : 8;   // nodes of the element per thread and sweep (N <= 8: 512 nodes = two per thread)
  __shared__ double s_tile[6][2][TS * LT];
  __shared__ double s_part[3][4][TS * LT];   // waves 1..3: their record's four lifted fields
  __shared__ double s_tr[TS * LT];
  __shared__ double s_Dfix[6][16];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int mi = lane & 15, mk = lane >> 4;
  for (int ei = blockIdx.x; ei < n_elem; ei += gridDim.x) {
    const int u0 = unit_first[ei], u1 = unit_first[ei + 1];
    if (u0 == u1) continue;
    const HangUnit first = units[u0];
    const ElemDesc el = ed[first.e];
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    const int KN = (N + 3) >> 2;
    // A u of the element, requested before anything else: it is added to at the very end
    double au_[NAU], rh_[FUSE ? NAU : 1], pp_[FUSE ? NAU : 1], uu_[FUSE ? NAU : 1];
    {
#pragma unroll
      for (int c = 0; c < NAU; ++c) {
        const bool in = threadIdx.x + c * TPB < N3;
        const size_t o = (size_t)el.ns + threadIdx.x + c * TPB;
        au_[c] = in ? Au[o] : 0.0;
        if constexpr (FUSE) { rh_[c] = in ? cf.rhs[o] : 0.0; pp_[c] = in ? cf.p[o] : 0.0; uu_[c] = in ? cf.u[o] : 0.0; }
      }
    }
    double opD[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) opD[ks] = (4 * ks + mk < N && mi < N) ? face_ops[el.offD + (4 * ks + mk) * N + mi] : 0.0;
    int mask = 0;
    for (int ui = u0; ui < u1; ++ui) {
      const HangUnit un = (ui == u0) ? first : units[ui];
      const int f = un.f, dir = f >> 1;
      const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;
      mask |= 1 << f;
      if (wv == 0 && lane < 16) s_Dfix[f][lane] = (lane < N) ? face_ops[el.offD + face_fix(f, N) * N + lane] : 0.0;
      mfma_d4 R[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c] = mfma_d4{0.0, 0.0, 0.0, 0.0};
      if (wv < un.nrec) {
        const HpMortar m = md[un.r0 + wv];
        const int NQ = m.NQ, T = NQ * NQ, KQ = (NQ + 3) >> 2;
        double oea[4], oeb[4];   // E along a / b: E[mi][4 ks + mk]  (N x NQ)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bool in = mi < N && 4 * ks + mk < NQ;
          oea[ks] = in ? hp_ops[m.offEa + mi * NQ + 4 * ks + mk] : 0.0;
          oeb[ks] = in ? hp_ops[m.offEb + mi * NQ + 4 * ks + mk] : 0.0;
        }
        double A[4][4];
        const bool robin = (m.kind == 0) && robin_c;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int bq = 4 * ks + mk;
          A[0][ks] = A[1][ks] = A[2][ks] = A[3][ks] = 0.0;
          if (mi < NQ && bq < NQ) {
            const int k = mi + NQ * bq;
            const double* qm = qtrace + m.qoff + k;
            const double um = qm[0];
            if (robin) {
              A[0][ks] = robin_c[m.gidx + k] * um - robin_r[m.gidx + k];
            } else {
              const double* g = geom + (size_t)7 * m.gidx + k;
              double up, tm = 0.0, tp = 0.0, am[3];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                am[i] = g[i * T];
                tm += am[i] * qm[(1 + i) * T];
              }
              if (m.kind != 0) {
                const double* pp = ((m.kind == 2) ? ghost_qtrace : qtrace) + m.nbr_qoff + reorder_index(m.code, NQ - 1, mi, bq);
                up = pp[m.u_shift];
#pragma unroll
                for (int i = 0; i < 3; ++i) tp += g[(3 + i) * T] * pp[(1 + i) * T];
              } else {
                up = bndry_q[m.gidx + k];
              }
              const double jump = um - up;
              const double w1 = (m.kind != 0) ? -0.5 : -1.0;
              A[0][ks] = w1 * (tm + tp) + g[6 * T] * jump;   // (fm, fp, w2: folded into the factors by face_geom_hp_kernel)
              A[1][ks] = w1 * am[t0] * jump;
              A[2][ks] = w1 * am[t1d] * jump;
              A[3][ks] = w1 * am[dir] * jump;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          mfma_d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (ks < KQ) y = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c][ks], oeb[ks], y, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < KQ) R[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(oea[r], y[r], R[c], 0, 0, 0);
        }
        if (wv > 0 && mi < TS) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (mk + 4 * r < TS) s_part[wv - 1][c][(mk + 4 * r) * LT + mi] = R[c][r];
        }
      }
      __syncthreads();
      if (wv == 0) {
        for (int w = 1; w < un.nrec; ++w) {   // the side's mortars, summed in record order
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (mi < TS && mk + 4 * r < TS) R[c][r] += s_part[w - 1][c][(mk + 4 * r) * LT + mi];
        }
        mfma_d4 val = R[0];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < KN) val = __builtin_amdgcn_mfma_f64_16x16x4f64(opD[r], R[1][r], val, 0, 0, 0);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mi < TS && mk + 4 * r < TS) s_tr[(mk + 4 * r) * LT + mi] = R[2][r];
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks >= KN) continue;
          const double a_ = (mi < TS && 4 * ks + mk < TS) ? s_tr[mi * LT + 4 * ks + mk] : 0.0;
          val = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, opD[ks], val, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mi < TS && mk + 4 * r < TS) {
            s_tile[f][0][(mk + 4 * r) * LT + mi] = val[r];
            s_tile[f][1][(mk + 4 * r) * LT + mi] = R[3][r];
          }
        }
      }
      __syncthreads();
    }
    for (int idx0 = threadIdx.x; idx0 < N3; idx0 += NAU * TPB) {
      if (idx0 != (int)threadIdx.x) {   // (N > 12: a second sweep)
#pragma unroll
        for (int c = 0; c < NAU; ++c) {
          const bool in = idx0 + c * TPB < N3;
          const size_t o = (size_t)el.ns + idx0 + c * TPB;
          au_[c] = in ? Au[o] : 0.0;
          if constexpr (FUSE) { rh_[c] = in ? cf.rhs[o] : 0.0; pp_[c] = in ? cf.p[o] : 0.0; uu_[c] = in ? cf.u[o] : 0.0; }
        }
      }
#pragma unroll
      for (int c = 0; c < NAU; ++c) {
        const int idx = idx0 + c * TPB;
        if (idx >= N3) continue;
        const int i = idx % N, j = (idx / N) % N, k = idx / N2;
        double v = 0.0;
        if (mask & 1) { v = fma(s_Dfix[0][i], s_tile[0][1][j * LT + k], v); if (i == 0) v += s_tile[0][0][j * LT + k]; }
        if (mask & 2) { v = fma(s_Dfix[1][i], s_tile[1][1][j * LT + k], v); if (i == N - 1) v += s_tile[1][0][j * LT + k]; }
        if (mask & 4) { v = fma(s_Dfix[2][j], s_tile[2][1][i * LT + k], v); if (j == 0) v += s_tile[2][0][i * LT + k]; }
        if (mask & 8) { v = fma(s_Dfix[3][j], s_tile[3][1][i * LT + k], v); if (j == N - 1) v += s_tile[3][0][i * LT + k]; }
        if (mask & 16) { v = fma(s_Dfix[4][k], s_tile[4][1][i * LT + j], v); if (k == 0) v += s_tile[4][0][i * LT + j]; }
        if (mask & 32) { v = fma(s_Dfix[5][k], s_tile[5][1][i * LT + j], v); if (k == N - 1) v += s_tile[5][0][i * LT + j]; }
        const double a_ = au_[c] + v;
        if (!FUSE || !cf.skip_Au_store) Au[el.ns + idx] = a_;
        if constexpr (FUSE) {
          const size_t o = (size_t)el.ns + idx;
          const double res = __dadd_rn(rh_[c], __dmul_rn(-1.0, a_));
          const double ri = __dmul_rn(cf.alpha, res);
          const double pi = __dadd_rn(__dmul_rn(cf.beta, pp_[c]), ri);
          if (cf.r) cf.r[o] = ri;
          cf.p[o] = pi;
          cf.u_out[o] = __dadd_rn(uu_[c], pi);
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

struct FaceHost {
  TraceUniform uni{};            // N == 0: descriptors are loaded
  // hanging-mesh (mortar record) path
  bool hp = false;
  int n_rec = 0;
  HpMortar* d_rec = nullptr;
  HpGeomSrc* d_gsrc = nullptr;
  int* d_elem_first = nullptr;
  int* d_side_first = nullptr;   // first record of side s (6 n_elements + 1 entries)
  int hp_max_N = 1, hp_max_NQ = 1;
  // hp split (deg, deg_quad <= 7): the fast conforming kernels serve every conforming (and small hanging) side of the mesh, the mortar-record
  // kernels only the hanging sides of the elements that have one (d_hang_elems)
  bool hp_split = false;
  bool hp_split_fast = false;    // ... and every degree <= 7: the p <= 7 kernels take every conforming side; else the tiled 16 x 16 kernels (or both
                                 // families on lists: family_split) do
  int* d_hang_elems = nullptr;
  int n_hang_elems = 0;
  HangUnit* d_units = nullptr;   // the sides of the listed elements that stay with the records, element by element (trace_unit_kernel / flux_unit_kernel)
  int* d_unit_first = nullptr;   // per listed element: its first unit (n_hang_elems + 1 entries)
  int n_units = 0;
  std::vector<HpMortar> rec_host;   // host copy of the records (set-up of the split)
  double* d_hp_ops = nullptr;
  int hp_fld_stride = 0;
  size_t hp_lds_doubles = 0;
  double* d_sj = nullptr;       // raw sj (for Robin data)
  double* d_robin_c = nullptr;  // sj * coeff, sj * rhs at the mortar nodes of the boundary sides
  double* d_robin_r = nullptr;
  bool robin = false;
  bool dirichlet_nonzero = false;   // non-zero Dirichlet data sits in d_bndry (it survives a Robin on / off cycle)
  std::vector<int> side_deg_m, side_deg_p;
  int* d_side_deg_m = nullptr;
  int* d_side_deg_p = nullptr;
  int* d_side_bndry_stride = nullptr;
  GhostSideDesc* d_ghost_sides = nullptr;
  int n_ghost_sides = 0;
  int fld_stride = 0;   // generic kernels: doubles per field buffer
  int max_N = 1, max_NQ = 1;
  int max_local_N = 1;   // largest deg + 1 among the plan's own elements (sizes the LDS copy of u in trace_mfma16_kernel)
  ElemDesc* d_elem_desc_generic = nullptr;  // offD -> unpadded N x N matrices
  // family split (conforming mixed-degree plans with degrees above 7): the elements whose own degree and six mortars all fit 8 x 8 go
  // through the p <= 7 kernels (trace_mfma_kernel / flux_wave_kernel, 2 - 3 x faster per element than the tiled 16 x 16 kernels), the
  // rest through the tiled ones; both families write the same trace array and add into the same A u
  bool family_split = false;
  int *d_fam_small = nullptr, *d_fam_big = nullptr;
  int n_fam_small = 0, n_fam_big = 0;
  // ... and the hybrid operator's ring / dirty lists of such a plan, split the same way (the list launches of the two families; without them
  // a listed launch sends every listed element through the tiled kernels: level 5, graded p = 3 ... 9, ring traces 396 us)
  const int *hy_ring = nullptr, *hy_dirty = nullptr;   // the hybrid operator's device lists these belong to (matched by pointer)
  const int* hy_dirty_any = nullptr;                   // the hybrid operator's dirty list, family split or not (a fused update may ride on it)
  int *d_ring_small = nullptr, *d_ring_big = nullptr, *d_dirty_small = nullptr, *d_dirty_big = nullptr;
  int n_ring_small = 0, n_ring_big = 0, n_dirty_small = 0, n_dirty_big = 0;
};
std::map<d4est_hip_plan*, FaceHost> g_face_host;

template <typename T>
T* upload_vec(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

}  // namespace

// dGMath/d4est_reference.c:3-12, :84-110: index, in the (+) side's own order, of the sub-face that is i in (-) order
int reorient_face_order(int f_m, int f_p, int o, int i) {
  using namespace topo;   // d4est_hip_topology.h: the tables, pinned entry by entry to the reference's own by tests/test_topology_tables.py
  return d4est_perm_to_order[d4est_code_to_perm[d4est_FToF_code[f_m][f_p]][o]][i];
}

// The (flip0, flip1, transpose) code d4est_operators_reorient_face_data derives from a tree-boundary face pair
// (dGMath/d4est_operators.c:2031-2050): p8est's face transform of the pair (lower face number, higher face number, orientation)
// -- axes of the lower face's tangential directions, the axes they map to, and whether each is reversed
// (p4est_expand_face_transform: my_axis = ft[0..2], target_axis = ft[3..5], edge_reverse = ft[6..8]).
int face_reorder_code(int f_m, int f_p, int o) {
  const auto& refs = topo::face_permutation_refs;
  const int* ref0 = refs[0];
  const int lo = f_m <= f_p ? f_m : f_p, hi = f_m <= f_p ? f_p : f_m;
  int my_axis[2], target_axis[2], edge_reverse[2];
  my_axis[0] = lo < 2 ? 1 : 0;
  my_axis[1] = lo < 4 ? 2 : 1;
  const int swap_axes = ref0[lo] ^ ref0[hi] ^ ((o == 0 || o == 3) ? 1 : 0);
  target_axis[swap_axes] = hi < 2 ? 1 : 0;
  target_axis[!swap_axes] = hi < 4 ? 2 : 1;
  const int swap_rev = (refs[lo][hi] == 1);
  edge_reverse[swap_rev] = o & 1;
  edge_reverse[!swap_rev] = o >> 1;
  const int aligned = (my_axis[1] - my_axis[0]) * (target_axis[1] - target_axis[0]) > 0;
  return edge_reverse[0] | (edge_reverse[1] << 1) | ((!aligned) << 2);
}

// Small side of a hanging face, sub-face c in its own (= (-)) order.  The reference slices the WHOLE big face, re-orients it with the
// (flip0, flip1, transpose) code and hp-prolongs the result onto child c (dGMath/d4est_laplacian_flux.c:575-657): u on sub-mortar c
// comes from the big element's own child that the code maps onto c.  The gradient comes from the child
// d4est_reference_reorient_face_order names (:733-830, :858-900).  The two agree whenever the code is the geometric map between the
// faces; for a transposed pair with exactly one flip seen from the lower-numbered face (codes 5, 6) the reference's re-orientation is
// not geometric (forest.reference_reorientation_is_consistent) and they differ -- followed here as the reference computes it.
static int small_side_u_child(int code, int c) {
  int ha = (code & 4) ? (c >> 1) : (c & 1), hb = (code & 4) ? (c & 1) : (c >> 1);
  if (code & 2) hb ^= 1;
  if (code & 1) ha ^= 1;
  return ha + 2 * hb;
}
static const char* kNonGeometricMsg =
    "the reference's re-orientation of this tree-face pair is not geometric (transposed with one flip, seen from the lower face)";

// Mortar records of a mesh with hanging faces (plan->side_hang etc. set by d4est_hip_plan_set_hanging).
static void faces_setup_hp(d4est_hip_plan* plan, FaceHost& fh) {
  const int ne = plan->n_elements;
  const int qt = plan->quad_type;
  std::vector<double> ops;
  std::map<std::tuple<int, int, int, int, int>, int> op_index;
  // C: side (deg_side) -> mortar quadrature nodes (deg_mq); child = -1: p-prolong, 0/1: hp-prolong onto that half
  auto get_C = [&](int deg_side, int deg_mq, int child) {
    auto key = std::make_tuple(0, deg_side, deg_mq, child, 0);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    const int nh = deg_mq + 1, nH = deg_side + 1;
    std::vector<double> P;
    if (child < 0) P = Tables1D::p_prolong(deg_side, deg_mq);
    else {
      std::vector<double> P2 = Tables1D::hp_prolong(deg_side, deg_mq);
      P.assign(P2.begin() + (size_t)child * nh * nH, P2.begin() + (size_t)(child + 1) * nh * nH);
    }
    std::vector<double> I = Tables1D::quad_interp(qt, deg_mq, deg_mq);
    std::vector<double> C = Tables1D::matmul(I, P, nh, nh, nH);
    const int off = (int)ops.size();
    ops.insert(ops.end(), C.begin(), C.end());
    op_index[key] = off;
    return off;
  };
  auto get_CD = [&](int deg_side, int deg_mq, int child) {
    auto key = std::make_tuple(2, deg_side, deg_mq, child, 0);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    const int offC = get_C(deg_side, deg_mq, child);
    std::vector<double> C(ops.begin() + offC, ops.begin() + offC + (size_t)(deg_mq + 1) * (deg_side + 1));
    std::vector<double> CD = Tables1D::matmul(C, Tables1D::dij(deg_side), deg_mq + 1, deg_side + 1, deg_side + 1);
    const int off = (int)ops.size();
    ops.insert(ops.end(), CD.begin(), CD.end());
    op_index[key] = off;
    return off;
  };
  // E: mortar quadrature nodes (deg_mq) -> Lobatto nodes of deg_ml (V^T W) -> side (p- or hp-prolong transposed)
  auto get_E = [&](int deg_m, int deg_ml, int deg_mq, int child) {
    auto key = std::make_tuple(1, deg_m, deg_ml, deg_mq, child);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    std::vector<double> I = Tables1D::quad_interp(qt, deg_ml, deg_mq);
    std::vector<double> w = Tables1D::quad_weights(qt, deg_mq);
    std::vector<double> ItW = Tables1D::transpose(I, deg_mq + 1, deg_ml + 1);
    for (int r = 0; r <= deg_ml; ++r)
      for (int c = 0; c <= deg_mq; ++c) ItW[(size_t)r * (deg_mq + 1) + c] *= w[c];
    const int nh = deg_ml + 1, nH = deg_m + 1;
    std::vector<double> P;
    if (child < 0) P = Tables1D::p_prolong(deg_m, deg_ml);
    else {
      std::vector<double> P2 = Tables1D::hp_prolong(deg_m, deg_ml);
      P.assign(P2.begin() + (size_t)child * nh * nH, P2.begin() + (size_t)(child + 1) * nh * nH);
    }
    std::vector<double> Pt = Tables1D::transpose(P, nh, nH);
    std::vector<double> E = Tables1D::matmul(Pt, ItW, nH, nh, deg_mq + 1);
    const int off = (int)ops.size();
    ops.insert(ops.end(), E.begin(), E.end());
    op_index[key] = off;
    return off;
  };
  const size_t ns = 6 * (size_t)ne;
  // element references: >= 0 local, <= -2 ghost g = -(ref + 2)
  auto valid_ref = [&](int ref) { return (ref >= 0 && ref < ne) || (ref <= -2 && -(ref + 2) < plan->n_ghost); };
  auto deg_of = [&](int ref) { return ref >= 0 ? plan->deg[ref] : plan->ghost_deg[-(ref + 2)]; };
  auto degq_of = [&](int ref) { return ref >= 0 ? plan->deg_quad[ref] : plan->ghost_deg_quad[-(ref + 2)]; };
  auto degq_mortar = [&](int em, int ep) { return std::max(degq_of(em), degq_of(ep)); };
  auto nodes2 = [](int deg) { return (deg + 1) * (deg + 1); };
  std::vector<HpMortar> rec;
  std::vector<HpGeomSrc> gsrc;
  std::vector<int> elem_first(ne + 1, 0), side_first(ns, 0);
  long long qoff = 0;
  int max_fld = 1, maxN = 1;
  for (int e = 0; e < ne; ++e) {
    elem_first[e] = (int)rec.size();
    const int deg_m = plan->deg[e], degq_m = plan->deg_quad[e];
    maxN = std::max(maxN, deg_m + 1);
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      side_first[s] = (int)rec.size();
      plan->trace_offset[s] = qoff;
      const int hang = plan->side_hang[s], nbr = plan->side_nbr[s], f_p = plan->side_nbr_face[s], o = plan->side_orientation[s];
      if (hang < 0 || hang > 2) D4EST_HIP_ABORT("plan_set_hanging: side %zu has side_hang %d", s, hang);
      if (o < 0 || o > 3) D4EST_HIP_ABORT("plan_set_hanging: side %zu has orientation %d", s, o);
      const int n_sub = (hang == 1) ? 4 : 1;
      const int* n4 = &plan->side_nbr4[4 * s];
      if (hang != 0)
        for (int i = 0; i < 4; ++i)
          if (!valid_ref(n4[i])) D4EST_HIP_ABORT("plan_set_hanging: side %zu: side_nbr4[%d] = %d is neither a local nor a ghost element", s, i, n4[i]);
      if (hang == 2 && !valid_ref(nbr)) D4EST_HIP_ABORT("plan_set_hanging: small side %zu: (+) element %d is neither local nor ghost", s, nbr);
      // mortar sizes of the whole hanging face, in (-) order and in (+) order
      int T_m[4] = {0, 0, 0, 0}, T_p[4] = {0, 0, 0, 0};
      if (hang != 0) {
        for (int i = 0; i < 4; ++i) {
          T_m[i] = nodes2(hang == 1 ? degq_mortar(e, n4[i]) : degq_mortar(n4[i], nbr));
          T_p[reorient_face_order(f, f_p, o, i)] = T_m[i];
        }
      }
      for (int i = 0; i < n_sub; ++i) {
        HpMortar m{};
        HpGeomSrc g{};
        m.elem = e;
        m.face = f;
        m.code = plan->side_reorder[s];
        m.N = deg_m + 1;
        m.first = (i == 0);
        m.last = (i == n_sub - 1);
        m.fm = m.fp = m.w2 = 1.0;
        m.hang = (double)hang;
        int ep = -1, sub_m = 0;   // (+) element of this mortar; index of the mortar in the face's (-) order
        if (hang == 0) {
          if (nbr != -1 && !valid_ref(nbr)) D4EST_HIP_ABORT("plan_set_faces: side %zu neighbour %d out of range", s, nbr);
          m.kind = (nbr == -1) ? 0 : 1;
          ep = nbr;
        } else if (hang == 1) {
          m.kind = 1;
          ep = n4[i];
          sub_m = i;
          m.fm = 0.5;   // the big element's gradient on the half-size mortar
          m.w2 = 0.5;
        } else {
          m.kind = 1;
          ep = nbr;
          sub_m = plan->side_sub[s];
          if (sub_m < 0 || sub_m > 3 || n4[sub_m] != e) D4EST_HIP_ABORT("plan_set_hanging: small side %zu: side_sub %d does not point at the element in side_nbr4", s, sub_m);
          m.fp = 0.5;
        }
        if (m.kind != 0 && ep <= -2) m.kind = 2;
        const int deg_p = (m.kind == 0) ? deg_m : deg_of(ep);
        const int deg_mq = (m.kind == 0) ? degq_m : degq_mortar(e, ep);
        const int deg_ml = std::max(deg_m, deg_p);
        m.NQ = deg_mq + 1;
        const int ca = (hang == 1) ? (i & 1) : -1, cb = (hang == 1) ? (i >> 1) : -1;
        m.offCa = get_C(deg_m, deg_mq, ca);
        m.offCb = get_C(deg_m, deg_mq, cb);
        m.offCDa = get_CD(deg_m, deg_mq, ca);
        m.offCDb = get_CD(deg_m, deg_mq, cb);
        m.offEa = get_E(deg_m, deg_ml, deg_mq, ca);
        m.offEb = get_E(deg_m, deg_ml, deg_mq, cb);
        g.S = plan->side_mortar_stride[s];
        g.off = 0;
        g.off_p = 0;
        g.Ttot = nodes2(deg_mq);
        if (hang != 0) {
          g.Ttot = T_m[0] + T_m[1] + T_m[2] + T_m[3];
          for (int j = 0; j < sub_m; ++j) g.off += T_m[j];
          const int sub_p = reorient_face_order(f, f_p, o, sub_m);
          for (int j = 0; j < sub_p; ++j) g.off_p += T_p[j];
        }
        g.deg_m = deg_m;
        g.deg_p = deg_p;
        m.gidx = g.S + g.off;
        if ((long long)m.gidx + nodes2(deg_mq) > plan->total_mortar_nodes) D4EST_HIP_ABORT("plan_set_faces: side %zu mortar data exceeds total_mortar_nodes", s);
        m.qoff = qoff;
        qoff += 4LL * nodes2(deg_mq);
        max_fld = std::max(max_fld, std::max(m.NQ * m.NQ, m.NQ * m.N));
        rec.push_back(m);
        gsrc.push_back(g);
      }
    }
  }
  elem_first[ne] = (int)rec.size();
  // (+) blocks: local ones by record lookup; ghost ones get consecutive slots of the ghost trace buffer in record order
  long long goff = 0;
  plan->rec_qoff.assign(rec.size(), 0);
  plan->rec_goff.assign(rec.size(), -1);
  plan->rec_len.assign(rec.size(), 0);
  plan->side_first_rec.assign(side_first.begin(), side_first.end());
  plan->side_first_rec.push_back((int)rec.size());
  for (size_t r = 0; r < rec.size(); ++r) {
    HpMortar& m = rec[r];
    plan->rec_qoff[r] = m.qoff;
    plan->rec_len[r] = 4 * m.NQ * m.NQ;
    if (m.kind == 2) {
      m.nbr_qoff = goff;
      plan->rec_goff[r] = goff;
      goff += 4LL * m.NQ * m.NQ;
      const size_t sg = 6 * (size_t)m.elem + m.face;
      if (plan->side_hang[sg] == 2 &&
          small_side_u_child(m.code, plan->side_sub[sg]) != reorient_face_order(m.face, plan->side_nbr_face[sg], plan->side_orientation[sg], plan->side_sub[sg]))
        D4EST_HIP_ABORT("plan_set_hanging: small side %zu: %s, and the big element is on another rank", sg, kNonGeometricMsg);
      continue;
    }
    if (m.kind == 0) continue;
    const size_t s = 6 * (size_t)m.elem + m.face;
    const int hang = plan->side_hang[s], f_p = plan->side_nbr_face[s], o = plan->side_orientation[s];
    int ep, sub_p = 0;
    if (hang == 1) ep = plan->side_nbr4[4 * s + (int)(r - side_first[s])];
    else ep = plan->side_nbr[s];
    const size_t sp = 6 * (size_t)ep + f_p;
    const int hang_p = plan->side_hang[sp];
    if ((hang == 0 && hang_p != 0) || (hang == 1 && hang_p != 2) || (hang == 2 && hang_p != 1))
      D4EST_HIP_ABORT("plan_set_hanging: sides %zu and %zu disagree about the hanging face", s, sp);
    if (hang == 2) sub_p = reorient_face_order(m.face, f_p, o, plan->side_sub[s]);   // the big element's own sub-mortar index
    const HpMortar& mp = rec[side_first[sp] + sub_p];
    if (mp.NQ != m.NQ) D4EST_HIP_ABORT("plan_set_hanging: sides %zu and %zu disagree on the mortar degree", s, sp);
    m.nbr_qoff = mp.qoff;
    if (hang == 2) {
      const int sub_u = small_side_u_child(m.code, plan->side_sub[s]);
      if (sub_u != sub_p) {
        const HpMortar& mu = rec[side_first[sp] + sub_u];
        if (mu.NQ != m.NQ) D4EST_HIP_ABORT("plan_set_hanging: small side %zu: %s, and its sub-mortars have different quadrature degrees", s, kNonGeometricMsg);
        m.u_shift = (int)(mu.qoff - mp.qoff);
      }
    }
  }
  plan->local_trace_doubles = qoff;
  plan->ghost_trace_doubles = goff;
  for (size_t s_ = 0; s_ < ns; ++s_) plan->ghost_trace_offset[s_] = plan->rec_goff[side_first[s_]];
  max_fld = std::max(max_fld, maxN * maxN);
  fh.hp = true;
  fh.n_rec = (int)rec.size();
  fh.hp_fld_stride = max_fld;
  fh.hp_lds_doubles = (size_t)12 * max_fld + (size_t)maxN * maxN * maxN + (size_t)maxN * maxN;
  if (fh.hp_lds_doubles * sizeof(double) > 160 * 1024) D4EST_HIP_ABORT("hanging-face kernels need %zu LDS doubles", fh.hp_lds_doubles);
  fh.d_rec = upload_vec(rec);
  fh.rec_host = rec;
  fh.d_gsrc = upload_vec(gsrc);
  fh.d_elem_first = upload_vec(elem_first);
  {
    std::vector<int> sf(side_first.begin(), side_first.end());
    sf.push_back((int)rec.size());
    fh.d_side_first = upload_vec(sf);
  }
  fh.hp_max_N = maxN;
  fh.hp_max_NQ = 1;
  for (const HpMortar& m_ : rec) fh.hp_max_NQ = std::max(fh.hp_max_NQ, m_.NQ);
  fh.d_hp_ops = upload_vec(ops);
  plan->face_fast = false;
}

void faces_setup(d4est_hip_plan* plan) {
  FaceHost& fh = g_face_host[plan];
  const int ne = plan->n_elements;
  const int qt = plan->quad_type;
  bool hp = false;
  if (!plan->side_hang.empty())
    for (int v : plan->side_hang) hp = hp || (v != 0);
  std::vector<double> ops;
  std::map<std::tuple<int, int, int, int>, int> op_index;  // (kind, deg_a, deg_b, deg_c) -> offset
  auto get_C = [&](int deg_side, int deg_mq) {
    auto key = std::make_tuple(0, deg_side, deg_mq, 0);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    // p-prolong to the Lobatto nodes of degree deg_mq, then Lobatto -> quadrature nodes (d4est_laplacian_flux.c:635-694)
    std::vector<double> P = Tables1D::p_prolong(deg_side, deg_mq);
    std::vector<double> I = Tables1D::quad_interp(qt, deg_mq, deg_mq);
    std::vector<double> C = Tables1D::matmul(I, P, deg_mq + 1, deg_mq + 1, deg_side + 1);
    const int off = (int)ops.size();
    ops.insert(ops.end(), C.begin(), C.end());
    op_index[key] = off;
    return off;
  };
  auto get_CD = [&](int deg_side, int deg_mq) {
    auto key = std::make_tuple(4, deg_side, deg_mq, 0);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    const int offC = get_C(deg_side, deg_mq);
    std::vector<double> C(ops.begin() + offC, ops.begin() + offC + (size_t)(deg_mq + 1) * (deg_side + 1));
    std::vector<double> CD = Tables1D::matmul(C, Tables1D::dij(deg_side), deg_mq + 1, deg_side + 1, deg_side + 1);
    const int off = (int)ops.size();
    ops.insert(ops.end(), CD.begin(), CD.end());
    op_index[key] = off;
    return off;
  };
  auto get_E = [&](int deg_m, int deg_ml, int deg_mq) {
    auto key = std::make_tuple(1, deg_m, deg_ml, deg_mq);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    // V^T W on the mortar (galerkin integral, deg_ml <- deg_mq) then P^T (deg_m <- deg_ml): sipg.c:641-734
    std::vector<double> I = Tables1D::quad_interp(qt, deg_ml, deg_mq);              // (mq+1) x (ml+1)
    std::vector<double> w = Tables1D::quad_weights(qt, deg_mq);
    std::vector<double> ItW = Tables1D::transpose(I, deg_mq + 1, deg_ml + 1);       // (ml+1) x (mq+1)
    for (int r = 0; r <= deg_ml; ++r)
      for (int c = 0; c <= deg_mq; ++c) ItW[(size_t)r * (deg_mq + 1) + c] *= w[c];
    std::vector<double> P = Tables1D::p_prolong(deg_m, deg_ml);                     // (ml+1) x (m+1)
    std::vector<double> Pt = Tables1D::transpose(P, deg_ml + 1, deg_m + 1);         // (m+1) x (ml+1)
    std::vector<double> E = Tables1D::matmul(Pt, ItW, deg_m + 1, deg_ml + 1, deg_mq + 1);
    const int off = (int)ops.size();
    ops.insert(ops.end(), E.begin(), E.end());
    op_index[key] = off;
    return off;
  };
  // derivative matrices: unpadded (generic kernels) and zero-padded 8 x 8 (fast path, degrees <= 7)
  auto get_D = [&](int deg, bool padded) {
    auto key = std::make_tuple(padded ? 3 : 2, deg, 0, 0);
    auto it = op_index.find(key);
    if (it != op_index.end()) return it->second;
    std::vector<double> D = Tables1D::dij(deg);
    const int n = deg + 1;
    const int off = (int)ops.size();
    if (!padded) {
      ops.insert(ops.end(), D.begin(), D.end());
    } else {
      std::vector<double> P(64, 0.0);
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) P[r * 8 + c] = D[(size_t)r * n + c];
      ops.insert(ops.end(), P.begin(), P.end());
    }
    op_index[key] = off;
    return off;
  };

  // per side sizes; the local trace block of side s is 4 T_s doubles in side order
  const size_t ns = 6 * (size_t)ne;
  std::vector<SideDesc> sd(ns);
  fh.side_deg_m.assign(ns, 0);
  fh.side_deg_p.assign(ns, 0);
  plan->trace_offset.assign(ns, 0);
  plan->ghost_trace_offset.assign(ns, -1);
  std::vector<GhostSideDesc> gsides;
  std::vector<long long> ghost_u_off(plan->n_ghost, 0);
  {
    long long o = 0;
    for (int g = 0; g < plan->n_ghost; ++g) {
      ghost_u_off[g] = o;
      const long long n = plan->ghost_deg[g] + 1;
      o += n * n * n;
    }
  }
  long long qoff = 0, goff = 0;
  // D4EST_HIP_TUNE_GHOST_ALIAS: every ghost side reads block 0 (a constant ghost trace, e.g. the zero trace of a Schwarz subdomain plan)
  const bool ghost_alias = plan->tuning[D4EST_HIP_TUNE_GHOST_ALIAS] > 0 && !hp;
  long long ghost_alias_len = 0;
  int maxN = 1, maxNQ = 1, max_fld = 1;
  bool fast = true;
  for (int e = 0; e < ne; ++e) maxN = std::max(maxN, plan->deg[e] + 1);
  for (int g = 0; g < plan->n_ghost; ++g) maxN = std::max(maxN, plan->ghost_deg[g] + 1);
  // first pass: mortar degrees and offsets
  std::vector<int> deg_mq_of(ns), deg_p_of(ns);
  for (int e = 0; e < ne; ++e)
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      const bool hanging = hp && plan->side_hang[s] != 0;
      // hanging sides are served by the mortar records (descriptor kind 3); a SMALL side still gets the degrees of its mortar with the
      // big element, so that the hp split can hand it to the conforming kernels (one mortar, p-prolongation only: it is one of theirs)
      const int nbr = hanging ? ((plan->side_hang[s] == 2 && plan->side_nbr[s] != -1) ? plan->side_nbr[s] : -1) : plan->side_nbr[s];
      const int deg_m = plan->deg[e], degq_m = plan->deg_quad[e];
      int deg_p = deg_m, degq_p = degq_m;
      if (nbr >= 0) {
        if (nbr >= ne) D4EST_HIP_ABORT("plan_set_faces: side %zu neighbour %d out of range", s, nbr);
        deg_p = plan->deg[nbr];
        degq_p = plan->deg_quad[nbr];
      } else if (nbr <= -2) {
        const int g = -(nbr + 2);
        if (g >= plan->n_ghost) D4EST_HIP_ABORT("plan_set_faces: side %zu ghost %d out of range", s, g);
        deg_p = plan->ghost_deg[g];
        degq_p = plan->ghost_deg_quad[g];
      }
      deg_mq_of[s] = (nbr == -1) ? degq_m : std::max(degq_m, degq_p);   // (max: also d4est's rule for a hanging face's mortars)
      deg_p_of[s] = deg_p;
      const long long T = (long long)(deg_mq_of[s] + 1) * (deg_mq_of[s] + 1);
      plan->trace_offset[s] = qoff;
      qoff += 4 * T;
      maxNQ = std::max(maxNQ, deg_mq_of[s] + 1);
    }
  plan->local_trace_doubles = qoff;
  fast = (maxN <= 8 && maxNQ <= 8);
  for (int e = 0; e < ne; ++e)
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      SideDesc d{};
      const bool hanging = hp && plan->side_hang[s] != 0;
      const int nbr = hanging ? -1 : plan->side_nbr[s];
      const int deg_m = plan->deg[e], deg_p = deg_p_of[s], deg_mq = deg_mq_of[s];
      const int deg_ml = std::max(deg_m, deg_p);
      d.kind = hanging ? 3 : ((nbr == -1) ? 0 : (nbr >= 0 ? 1 : 2));
      d.code = plan->side_reorder[s];
      d.NQ = deg_mq + 1;
      d.offC = get_C(deg_m, deg_mq);
      d.offCD = get_CD(deg_m, deg_mq);
      d.pad0 = 0;
      d.offE = get_E(deg_m, d.kind == 0 ? deg_m : deg_ml, deg_mq);
      d.geom = plan->side_mortar_stride[s];
      d.qoff = plan->trace_offset[s];
      d.nbr_qoff = 0;
      if (d.kind == 1 && !hp) {
        const size_t sp = 6 * (size_t)nbr + plan->side_nbr_face[s];
        if (deg_mq_of[sp] != deg_mq) D4EST_HIP_ABORT("plan_set_faces: sides %zu and %zu disagree on the mortar degree (non-conforming mortar?)", s, sp);
        d.nbr_qoff = plan->trace_offset[sp];
      } else if (d.kind == 2 && !hp) {
        const int g = -(nbr + 2);
        plan->ghost_trace_offset[s] = goff;
        d.nbr_qoff = goff;
        GhostSideDesc gs{};
        gs.N = plan->ghost_deg[g] + 1;
        gs.f = plan->side_nbr_face[s];
        gs.NQ = deg_mq + 1;
        gs.offC = get_C(plan->ghost_deg[g], deg_mq);
        gs.offD = get_D(plan->ghost_deg[g], false);
        gs.u_off = ghost_u_off[g];
        gs.goff = goff;
        gsides.push_back(gs);
        if (ghost_alias) ghost_alias_len = std::max(ghost_alias_len, 4LL * (deg_mq + 1) * (deg_mq + 1));
        else goff += 4LL * (deg_mq + 1) * (deg_mq + 1);
      }
      sd[s] = d;
      fh.side_deg_m[s] = deg_m;
      fh.side_deg_p[s] = deg_p;
      max_fld = std::max(max_fld, std::max(d.NQ * d.NQ, d.NQ * (deg_m + 1)));
      max_fld = std::max(max_fld, d.NQ * (deg_p + 1));
    }
  plan->ghost_trace_doubles = ghost_alias ? ghost_alias_len : goff;   // (hanging plans: overwritten by faces_setup_hp)
  max_fld = std::max(max_fld, maxN * maxN);
  fh.fld_stride = max_fld;
  fh.max_N = maxN;
  fh.max_NQ = maxNQ;
  fh.max_local_N = 1;
  for (int e = 0; e < ne; ++e) fh.max_local_N = std::max(fh.max_local_N, plan->deg[e] + 1);
  fh.n_ghost_sides = (int)gsides.size();
  plan->face_fast = fast;
  plan->max_face_lds_doubles = 12 * max_fld + maxN * maxN * maxN + maxN * maxN;
  if ((size_t)plan->max_face_lds_doubles * sizeof(double) > 160 * 1024) D4EST_HIP_ABORT("face kernels need %d LDS doubles", plan->max_face_lds_doubles);

  std::vector<ElemDesc> edv(ne), edg(ne);
  for (int e = 0; e < ne; ++e) {
    edv[e].N = edg[e].N = plan->deg[e] + 1;
    edv[e].ns = edg[e].ns = plan->nodal_stride[e];
    edg[e].offD = get_D(plan->deg[e], false);
    edv[e].offD = (fast || plan->deg[e] + 1 <= 8) ? get_D(plan->deg[e], true) : edg[e].offD;
    edv[e].pad = edg[e].pad = 0;
  }
  // uniform plan? (one degree, one mortar degree, contiguous element and trace strides)
  fh.uni = TraceUniform{};
  TraceUniform un{};
  if (ne > 0 && !hp) {
    bool uniform = true;
    const int N0 = edv[0].N, n3 = N0 * N0 * N0;
    for (int e = 0; e < ne && uniform; ++e) uniform = (edv[e].N == N0) && (edv[e].ns == edv[0].ns + e * n3);
    const long long qs = 4LL * sd[0].NQ * sd[0].NQ;
    for (size_t s_ = 0; s_ < ns && uniform; ++s_)
      uniform = (sd[s_].NQ == sd[0].NQ) && (sd[s_].offC == sd[0].offC) && (sd[s_].offCD == sd[0].offCD) && (sd[s_].qoff == sd[0].qoff + (long long)s_ * qs);
    if (uniform) {
      un.N = N0; un.NQ = sd[0].NQ; un.offC = sd[0].offC; un.offCD = sd[0].offCD; un.offD = edv[0].offD;
      un.ns0 = edv[0].ns; un.ns_stride = n3; un.q0 = sd[0].qoff; un.q_stride = qs;
    }
  }
  if (fast) fh.uni = un;   // (the uniform forms of the trace kernels exist for N, NQ <= 8 only)
  // the direct kernels (d4est_hip_direct.hip: one wavefront per element, deg_quad <= 7; d4est_hip_direct_mw.hip: one multi-wave
  // workgroup per element, deg = deg_quad = 8 ... 15) take over apply_aij on uniform conforming plans whose sides all see the local degree
  direct_destroy(plan);
  if (un.N > 0) {
    bool same = true;
    for (size_t s_ = 0; s_ < ns && same; ++s_) same = (sd[s_].offE == sd[0].offE) && (deg_p_of[s_] == plan->deg[0]) && sd[s_].kind != 3;
    if (same && sd[0].NQ >= un.N && (fast ? sd[0].NQ * sd[0].NQ <= 64 : true))
      direct_setup(plan, un.N, sd[0].NQ, un.ns0, un.ns_stride, ops.data() + un.offC, ops.data() + un.offCD, ops.data() + sd[0].offE);
  }
  ops.resize(ops.size() + 64, 0.0);  // slack: the fast kernels read 64-entry images
  plan->d_elem_desc = upload_vec(edv);
  fh.d_elem_desc_generic = upload_vec(edg);
  HIP_CHECK(hipMalloc(&plan->d_side_desc, std::max<size_t>(sd.size(), 1) * sizeof(SideDesc)));
  if (!sd.empty()) HIP_CHECK(hipMemcpy(plan->d_side_desc, sd.data(), sd.size() * sizeof(SideDesc), hipMemcpyHostToDevice));
  plan->d_face_ops = upload_vec(ops);
  fh.d_side_deg_m = upload_vec(fh.side_deg_m);
  fh.d_side_deg_p = upload_vec(fh.side_deg_p);
  fh.d_side_bndry_stride = upload_vec(plan->side_bndry_stride);
  fh.d_ghost_sides = upload_vec(gsides);
  if (hp) faces_setup_hp(plan, fh);
  // ---- hp split.  A mesh with hanging faces is mostly conforming (2:1 interfaces of a locally refined forest): where every degree is
  // <= 7 the conforming sides go to the fast kernels of the conforming path (one wavefront per face, scalar-operand contractions) and
  // only the hanging sides to the tiled mortar-record kernels, over the list of elements that have one.  Both families write the same
  // trace array (offsets of the records: a conforming side's block is its single record's) and add into the same A u.  A hanging
  // side keeps kind 3 in the conforming descriptors: the fast trace kernel fills its block with a pretend-conforming trace that the
  // record kernel, launched after it, overwrites (the block is at least as long), the fast flux kernel reads zeros for it.
  fh.hp_split = false;
  (void)hipFree(fh.d_hang_elems); fh.d_hang_elems = nullptr; fh.n_hang_elems = 0;
  (void)hipFree(fh.d_units); (void)hipFree(fh.d_unit_first); fh.d_units = nullptr; fh.d_unit_first = nullptr; fh.n_units = 0;
  // (degrees above 7 -- round 4: the conforming sides then go through the tiled 16 x 16 conforming kernels, or, family split below, through
  // both conforming families on lists; D4EST_HIP_HP_SPLIT_FAST_ONLY=1 keeps such plans on the record kernels throughout)
  const bool split_fast = fast && fh.hp_max_N <= 8 && fh.hp_max_NQ <= 8;
  const bool split_tiled = !split_fast && fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16 && fh.max_N <= 16 && fh.max_NQ <= 16 && !std::getenv("D4EST_HIP_HP_SPLIT_FAST_ONLY");
  fh.hp_split_fast = false;
  if (hp && (split_fast || split_tiled) && plan->tuning[D4EST_HIP_TUNE_HP_SPLIT] != 0 &&
      plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 1) {
    std::vector<int> hang_elems;
    std::vector<HpMortar> rec = fh.rec_host;
    bool ok = true;
    for (int e = 0; e < ne && ok; ++e) {
      bool any = false;
      for (int f = 0; f < 6; ++f) {
        const size_t s_ = 6 * (size_t)e + f;
        SideDesc& d = sd[s_];
        d.qoff = plan->trace_offset[s_];
        const int hang = plan->side_hang[s_];
        if (hang == 2) {
          // a small side: ONE mortar, its own trace p-prolonged to the mortar nodes, the (+) block = the big element's sub-mortar, the
          // hanging factor folded into the geometric factors -- to the conforming kernels it is an ordinary interior side (unless
          // the reference's non-geometric re-orientation shifts the block it reads u from: those stay with the record kernel)
          HpMortar& m = rec[plan->side_first_rec[s_]];
          if ((m.kind == 1 || m.kind == 2) && m.u_shift == 0 && m.NQ == d.NQ && plan->side_first_rec[s_ + 1] == plan->side_first_rec[s_] + 1) {
            d.kind = m.kind;   // (2: the big element is a ghost -- its sub-mortar block arrives in the ghost trace buffer at the record's offset)
            d.geom = m.gidx;
            d.nbr_qoff = m.nbr_qoff;
            m.hang = 0.0;   // (the record kernel skips it)
            continue;
          }
        }
        if (hang != 0) { any = true; continue; }
        if (d.kind == 1) {
          const size_t sp = 6 * (size_t)plan->side_nbr[s_] + plan->side_nbr_face[s_];
          if (plan->side_hang[sp] != 0 || deg_mq_of[sp] != deg_mq_of[s_]) { ok = false; break; }
          d.nbr_qoff = plan->trace_offset[sp];
        } else if (d.kind == 2) {
          // a conforming side against a ghost: the exchanged block sits where the side's (single) record expects it
          const HpMortar& m = rec[plan->side_first_rec[s_]];
          if (m.kind != 2 || m.NQ != d.NQ || m.u_shift != 0) { ok = false; break; }
          d.nbr_qoff = m.nbr_qoff;
        }
      }
      if (any) hang_elems.push_back(e);
    }
    // measured (level-4 brick, p = 7, every 64th / 32nd / 16th / 8th / 3rd octant refined): apply_aij 203 -> 139, 236 -> 170, 292 -> 210,
    // 402 -> 292, 661 -> 448 us.  The record kernel's cost is per listed element (those with a BIG hanging side, or a small side the
    // conforming kernels cannot take): should they ever be more than half of the mesh, the split is not taken (tuning value 1 forces it)
    if (ok && plan->tuning[D4EST_HIP_TUNE_HP_SPLIT] < 0 && 2 * hang_elems.size() > (size_t)ne) ok = false;
    if (ok) {
      fh.hp_split = true;
      fh.hp_split_fast = split_fast;
      fh.n_hang_elems = (int)hang_elems.size();
      fh.d_hang_elems = upload_vec(hang_elems);
      // the units: per listed element its sides with a live record (D4EST_HIP_NO_HANG_UNITS=1 keeps the serial record kernels)
      if (!std::getenv("D4EST_HIP_NO_HANG_UNITS")) {
        std::vector<HangUnit> units;
        std::vector<int> unit_first(1, 0);
        bool units_ok = true;
        for (int e : hang_elems) {
          for (int f = 0; f < 6; ++f) {
            const size_t s_ = 6 * (size_t)e + f;
            const int r0 = plan->side_first_rec[s_], r1 = plan->side_first_rec[s_ + 1];
            int live = 0;
            for (int r = r0; r < r1; ++r) live += rec[r].hang != 0.0;
            if (live == 0) continue;
            if (live != r1 - r0 || r1 - r0 > 4) { units_ok = false; break; }
            units.push_back(HangUnit{e, f, r0, r1 - r0});
          }
          unit_first.push_back((int)units.size());
        }
        if (units_ok && !units.empty()) {
          fh.n_units = (int)units.size();
          fh.d_units = upload_vec(units);
          fh.d_unit_first = upload_vec(unit_first);
        }
      }
      if (!sd.empty()) HIP_CHECK(hipMemcpy(plan->d_side_desc, sd.data(), sd.size() * sizeof(SideDesc), hipMemcpyHostToDevice));
      if (!rec.empty()) HIP_CHECK(hipMemcpy(fh.d_rec, rec.data(), rec.size() * sizeof(HpMortar), hipMemcpyHostToDevice));
    }
  }
  // ---- family split of a conforming mixed-degree plan whose largest degree is above 7 (see FaceHost)
  fh.family_split = false;
  (void)hipFree(fh.d_fam_small); (void)hipFree(fh.d_fam_big);
  fh.d_fam_small = fh.d_fam_big = nullptr; fh.n_fam_small = fh.n_fam_big = 0;
  if ((!hp || (fh.hp_split && !fh.hp_split_fast)) && !fast && ne > 0 && fh.max_N <= 16 && fh.max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 &&
      plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 1 && plan->tuning[D4EST_HIP_TUNE_GHOST_ALIAS] <= 0 && !std::getenv("D4EST_HIP_NO_FAMILY_SPLIT")) {
    std::vector<int> small_e, big_e;
    for (int e = 0; e < ne; ++e) {
      bool sm = plan->deg[e] + 1 <= 8;
      for (int f = 0; f < 6 && sm; ++f) {
        const size_t s_ = 6 * (size_t)e + f;
        sm = deg_mq_of[s_] + 1 <= 8 && deg_p_of[s_] + 1 <= 8;
      }
      (sm ? small_e : big_e).push_back(e);
    }
    if (!big_e.empty() && 2 * small_e.size() >= (size_t)ne) {
      fh.family_split = true;
      fh.d_fam_small = upload_vec(small_e); fh.n_fam_small = (int)small_e.size();
      fh.d_fam_big = upload_vec(big_e); fh.n_fam_big = (int)big_e.size();
    }
  }
  // ---- the hybrid operator (d4est_hip_direct.hip): on mixed-degree / locally refined plans the CLEAN elements -- deg_quad = deg with a
  // one-kernel instance, all six sides conforming against a local element of the same degree or the boundary -- take the trace-free
  // whole-operator kernels of their degree bucket; one rank, plans whose two-phase kernels have list forms
  hybrid_destroy(plan);
  (void)hipFree(fh.d_ring_small); (void)hipFree(fh.d_ring_big); (void)hipFree(fh.d_dirty_small); (void)hipFree(fh.d_dirty_big);
  fh.d_ring_small = fh.d_ring_big = fh.d_dirty_small = fh.d_dirty_big = nullptr;
  fh.n_ring_small = fh.n_ring_big = fh.n_dirty_small = fh.n_dirty_big = 0;
  fh.hy_ring = fh.hy_dirty = fh.hy_dirty_any = nullptr;
  if (!plan->direct && ne > 0 && plan->n_ghost == 0 && plan->tuning[D4EST_HIP_TUNE_HYBRID] != 0 && plan->tuning[D4EST_HIP_TUNE_GHOST_ALIAS] <= 0 &&
      plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 1 && (hp ? (fh.hp_split || (fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16)) : (fast || (fh.max_N <= 16 && fh.max_NQ <= 16)))) {
    std::vector<char> bucket_ok(plan->buckets.size(), 0), clean(ne, 0);
    std::vector<int> bucket_of(ne, -1);
    // hanging-aware form (hp split active): a hanging side does not make an element dirty -- the record kernels serve it (kind 3) or,
    // a small side the split handed to the conforming kernels, the direct kernel reads the big element's sub-mortar block from the trace
    // array and exports its own (kind 2); D4EST_HIP_HYBRID_NO_HANGING=1 keeps every element with a hanging side dirty
    // (plans with degrees above 7 under the generalised hp split: the elements of the one-wavefront buckets, N <= 8, only -- the multi-wave
    // whole-operator kernel has no hanging-aware instance)
    const bool hang_aware = hp && fh.hp_split && !std::getenv("D4EST_HIP_HYBRID_NO_HANGING") && plan->local_trace_doubles < (1LL << 31);
    // mixed-aware form: a conforming side against a local element of LOWER degree does not make the (one-wavefront) element dirty either --
    // the mortar is the element's own (d4est's rule: the larger degree), its operators are the same-degree ones, and the (+) block, which the
    // lower-degree neighbour's trace kernel interpolates up to that mortar, is read from the trace array like a ghost block (kind 2; the
    // neighbour itself stays dirty: ITS side sees a mortar above its own degree).  D4EST_HIP_HYBRID_NO_MIXED=1 switches it off
    const bool mixed_aware = (!hp || fh.hp_split) && !std::getenv("D4EST_HIP_HYBRID_NO_MIXED") && plan->local_trace_doubles < (1LL << 31);
    std::vector<HybridSideOverride> ov;
    if (hang_aware || mixed_aware) ov.assign(ns, HybridSideOverride{-1, 0, 0, 0});
    bool any_ov = false, any_hang_ov = false, any_mixed_ov = false;
    for (size_t b = 0; b < plan->buckets.size(); ++b) {
      const Bucket& bk = plan->buckets[b];
      bucket_ok[b] = bk.N == bk.NQ && bk.d_EBf && hybrid_pair_built(bk.N, bk.NQ);
      for (int i = 0; i < bk.n_elem; ++i) bucket_of[plan->elem_ids[bk.elem_offset + i]] = (int)b;
    }
    int n_clean = 0;
    for (int e = 0; e < ne; ++e) {
      if (bucket_of[e] < 0 || !bucket_ok[bucket_of[e]]) continue;
      bool ok = true;
      for (int f = 0; f < 6 && ok; ++f) {
        const size_t s_ = 6 * (size_t)e + f;
        if (hp && plan->side_hang[s_] != 0) {
          if (!hang_aware || plan->buckets[bucket_of[e]].N > 8) { ok = false; break; }
          const SideDesc& d = sd[s_];
          if (d.kind == 3) { ov[s_] = HybridSideOverride{3, 0, 0, 0}; continue; }   // (the element is on the record kernels' list)
          if (d.kind == 1 && plan->side_hang[s_] == 2 && d.NQ == plan->buckets[bucket_of[e]].NQ && deg_p_of[s_] == plan->deg[e]) {
            ov[s_] = HybridSideOverride{2, d.nbr_qoff, (int)d.qoff, d.geom};
            continue;
          }
          ok = false;
          break;
        }
        const int nbr = plan->side_nbr[s_];
        if (nbr == -1) continue;                                   // domain boundary
        if (nbr < 0) { ok = false; break; }                        // (ghost: not on one-rank plans)
        const size_t sp = 6 * (size_t)nbr + plan->side_nbr_face[s_];
        if (mixed_aware && plan->buckets[bucket_of[e]].N <= 8 && plan->deg[nbr] < plan->deg[e] && plan->deg_quad[nbr] <= plan->deg_quad[e] &&
            deg_mq_of[s_] == plan->deg_quad[e] && !(hp && plan->side_hang[sp] != 0) && sd[s_].kind == 1 && sd[s_].NQ == plan->buckets[bucket_of[e]].NQ) {
          // (no export: the neighbour is dirty, so this element is in the ring and the trace kernel writes its block -- an export from the
          // clean kernel could race with the dirty flux kernel reading that block on another stream)
          ov[s_] = HybridSideOverride{2, sd[s_].nbr_qoff, -1, sd[s_].geom};
          continue;
        }
        ok = plan->deg[nbr] == plan->deg[e] && plan->deg_quad[nbr] == plan->deg_quad[e] && deg_mq_of[s_] == plan->deg_quad[e] &&
             !(hp && plan->side_hang[sp] != 0);
      }
      clean[e] = ok;
      n_clean += ok;
      if (ok && (hang_aware || mixed_aware))
        for (int f = 0; f < 6; ++f) {
          const size_t s_ = 6 * (size_t)e + f;
          if (ov[s_].kind < 0) continue;
          any_ov = true;
          if (hp && plan->side_hang[s_] != 0) any_hang_ov = true;
          else any_mixed_ov = true;
        }
    }
    // default: only where it was measured to pay -- ONE clean degree bucket holding at least half of the elements (a locally refined mesh of
    // one degree: level 4, p = 7, every 64th octant refined 144 -> 125 us).  With several clean buckets every bucket is its own launch of
    // a latency-structured kernel (25 - 40 us for a few hundred elements each, whatever their number): back to back they cost more than
    // the two-phase kernels save, and on side streams the cross-queue event waits (about 50 us per fork / join on this runtime) eat the
    // overlap (graded p = 3 ... 9, 4096 elements: 144 us two-phase, 275 us serial, 250 us forked) -- tuning value 1 still forces it
    // (round 4: with several clean buckets the default keeps the LARGEST one where it holds at least half of the mesh -- a mesh with one
    // dominant degree -- and leaves the other buckets' elements to the two-phase lists)
    int clean_buckets = 0;
    bool all_buckets = false;
    {
      std::vector<int> cnt(plan->buckets.size(), 0);
      for (int e = 0; e < ne; ++e)
        if (clean[e]) ++cnt[bucket_of[e]];
      int best = -1;
      for (size_t b = 0; b < cnt.size(); ++b) {
        if (cnt[b] > 0) ++clean_buckets;
        if (best < 0 || cnt[b] > cnt[best]) best = (int)b;
      }
      // ... and, at size, every clean bucket of at least 2048 clean elements (the smaller buckets' elements stay two-phase): with thousands of clean elements per bucket the buckets' launches fill the chip and their
      // latency no longer matters (level 5, 32 768 elements, graded p = 3 ... 9: two-phase 756 us, hybrid 691 us; hanging + plateaus 1033 -> 894 us;
      // at level 4, 385 clean elements per bucket: 275 against 126 us)
      const bool one_only = std::getenv("D4EST_HIP_HYBRID_ONE_BUCKET_ONLY") != nullptr;
      if (std::getenv("D4EST_HIP_DEBUG_HYBRID")) {
        std::fprintf(stderr, "[d4est_hip] hybrid classification: %d of %d elements clean, per bucket:", n_clean, ne);
        for (size_t b = 0; b < cnt.size(); ++b) std::fprintf(stderr, " N=%d:%d/%d", plan->buckets[b].N, cnt[b], plan->buckets[b].n_elem);
        std::fprintf(stderr, "\n");
      }
      // the buckets that are worth a launch of their own at size: at least 2048 clean elements each
      int n_kept = 0;
      size_t kept_sum = 0;
      for (size_t b = 0; b < cnt.size(); ++b)
        if (cnt[b] >= 2048) { ++n_kept; kept_sum += (size_t)cnt[b]; }
      if (plan->tuning[D4EST_HIP_TUNE_HYBRID] < 0 && clean_buckets > 1 && !one_only && n_kept >= 2 && 2 * kept_sum >= (size_t)ne) {
        for (int e = 0; e < ne; ++e)
          if (clean[e] && cnt[bucket_of[e]] < 2048) { clean[e] = 0; --n_clean; }   // (small buckets' elements: two-phase)
        all_buckets = true;
      } else if (plan->tuning[D4EST_HIP_TUNE_HYBRID] < 0 && clean_buckets > 1 && best >= 0 && 2 * (size_t)cnt[best] >= (size_t)ne && !one_only) {
        for (int e = 0; e < ne; ++e)
          if (clean[e] && bucket_of[e] != best) { clean[e] = 0; --n_clean; }
        clean_buckets = 1;
      }
    }
    if (n_clean > 0 && (plan->tuning[D4EST_HIP_TUNE_HYBRID] > 0 || all_buckets || (clean_buckets == 1 && 2 * (size_t)n_clean >= (size_t)ne))) {
      std::vector<int> oC(plan->buckets.size(), -1), oCD(plan->buckets.size(), -1), oE(plan->buckets.size(), -1);
      for (size_t b = 0; b < plan->buckets.size(); ++b) {
        if (!bucket_ok[b]) continue;
        const int d = plan->buckets[b].deg, dq = plan->buckets[b].deg_quad;
        oC[b] = get_C(d, dq); oCD[b] = get_CD(d, dq); oE[b] = get_E(d, d, dq);
      }
      std::vector<const double*> pC(plan->buckets.size(), nullptr), pCD(plan->buckets.size(), nullptr), pE(plan->buckets.size(), nullptr);
      for (size_t b = 0; b < plan->buckets.size(); ++b)
        if (oC[b] >= 0) { pC[b] = ops.data() + oC[b]; pCD[b] = ops.data() + oCD[b]; pE[b] = ops.data() + oE[b]; }
      // (the flags describe the classification before the dominant-bucket rule demoted anybody: a label, not a contract)
      hybrid_setup(plan, clean, pC, pCD, pE, any_ov ? &ov : nullptr,
                   any_hang_ov ? (any_mixed_ov ? "hanging-aware, mixed-aware" : "hanging-aware") : "mixed-aware");
      {
        const int *dd0, *dr0;
        int nd0, nr0;
        hybrid_lists(plan, &dd0, &nd0, &dr0, &nr0);
        fh.hy_dirty_any = dd0;
      }
      if (fh.family_split) {
        std::vector<char> is_small(ne, 0);
        for (int e = 0; e < ne; ++e) {
          bool sm = plan->deg[e] + 1 <= 8;
          for (int f = 0; f < 6 && sm; ++f) {
            const size_t s_ = 6 * (size_t)e + f;
            sm = deg_mq_of[s_] + 1 <= 8 && deg_p_of[s_] + 1 <= 8;
          }
          is_small[e] = sm;
        }
        const std::vector<int>*hd, *hr;
        hybrid_host_lists(plan, &hd, &hr);
        std::vector<int> rs, rb, ds, db;
        for (int e : *hr) (is_small[e] ? rs : rb).push_back(e);
        for (int e : *hd) (is_small[e] ? ds : db).push_back(e);
        fh.d_ring_small = upload_vec(rs); fh.n_ring_small = (int)rs.size();
        fh.d_ring_big = upload_vec(rb); fh.n_ring_big = (int)rb.size();
        fh.d_dirty_small = upload_vec(ds); fh.n_dirty_small = (int)ds.size();
        fh.d_dirty_big = upload_vec(db); fh.n_dirty_big = (int)db.size();
        const int *dd, *dr;
        int nd_, nr_;
        hybrid_lists(plan, &dd, &nd_, &dr, &nr_);
        fh.hy_dirty = dd; fh.hy_ring = dr;
      }
    }
  }
  const size_t tm = std::max<size_t>((size_t)plan->total_mortar_nodes, 1);
  HIP_CHECK(hipMalloc(&plan->d_trace, std::max<size_t>((size_t)plan->local_trace_doubles, 1) * sizeof(double)));
  HIP_CHECK(hipMalloc(&plan->d_bndry, tm * sizeof(double)));  // Dirichlet data at the mortar quadrature nodes, by geom stride
  HIP_CHECK(hipMemset(plan->d_bndry, 0, tm * sizeof(double)));
  HIP_CHECK(hipMalloc(&plan->d_face_geom, 7 * tm * sizeof(double)));
  plan->has_faces = true;
}

void faces_set_geometry(d4est_hip_plan* plan, const double* sj, const double* n, const double* drst_m, const double* drst_p,
                        const double* hm, const double* hp, int on_device) {
  FaceHost& fh = g_face_host[plan];
  const size_t T = (size_t)plan->total_mortar_nodes;
  const double* src[6] = {sj, n, drst_m, drst_p, hm, hp};
  const size_t mult[6] = {1, 3, 9, 9, 1, 1};
  double* tmp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const double* dev[6];
  for (int i = 0; i < 6; ++i) {
    if (!src[i]) D4EST_HIP_ABORT("plan_set_mortar_geometry: NULL array %d", i);
    if (on_device) {
      dev[i] = src[i];
    } else {
      HIP_CHECK(hipMalloc(&tmp[i], std::max<size_t>(mult[i] * T, 1) * sizeof(double)));
      HIP_CHECK(hipMemcpy(tmp[i], src[i], mult[i] * T * sizeof(double), hipMemcpyHostToDevice));
      dev[i] = tmp[i];
    }
  }
  const int n_sides = 6 * plan->n_elements;
  if (fh.hp) {
    if (fh.n_rec > 0)
      hipLaunchKernelGGL(face_geom_hp_kernel, dim3(std::min(fh.n_rec, 8192)), dim3(64), 0, plan->stream, fh.d_rec, fh.d_gsrc, fh.n_rec,
                         dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], plan->sipg_prefactor, plan->sipg_penalty_fcn, plan->d_face_geom);
    HIP_CHECK(hipGetLastError());
  } else if (n_sides > 0) {
    const int grid = n_sides < 8192 ? n_sides : 8192;
    hipLaunchKernelGGL(face_geom_kernel, dim3(grid), dim3(64), 0, plan->stream, (const SideDesc*)plan->d_side_desc,
                       fh.d_side_deg_m, fh.d_side_deg_p, n_sides, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5],
                       plan->sipg_prefactor, plan->sipg_penalty_fcn, plan->d_face_geom);
    HIP_CHECK(hipGetLastError());
  }
  // raw sj is kept for Robin boundary data (faces_set_robin)
  if (!fh.d_sj) HIP_CHECK(hipMalloc(&fh.d_sj, std::max<size_t>(T, 1) * sizeof(double)));
  if (T > 0) HIP_CHECK(hipMemcpyAsync(fh.d_sj, dev[0], T * sizeof(double), hipMemcpyDeviceToDevice, plan->stream));
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  for (int i = 0; i < 6; ++i)
    if (tmp[i]) HIP_CHECK(hipFree(tmp[i]));
  plan->has_face_geometry = true;
}

void faces_set_dirichlet(d4est_hip_plan* plan, const double* g_lobatto, int on_device) {
  FaceHost& fh = g_face_host[plan];
  const size_t tm = std::max<size_t>((size_t)plan->total_mortar_nodes, 1);
  fh.dirichlet_nonzero = (g_lobatto != nullptr);
  plan->bc_inhomogeneous = fh.dirichlet_nonzero || fh.robin;
  if (!g_lobatto) {
    HIP_CHECK(hipMemsetAsync(plan->d_bndry, 0, tm * sizeof(double), plan->stream));
    return;
  }
  const size_t nb = (size_t)plan->total_bndry_nodes;
  if (nb == 0) return;
  const double* dev = g_lobatto;
  double* tmp = nullptr;
  if (!on_device) {
    HIP_CHECK(hipMalloc(&tmp, nb * sizeof(double)));
    HIP_CHECK(hipMemcpy(tmp, g_lobatto, nb * sizeof(double), hipMemcpyHostToDevice));
    dev = tmp;
  }
  const int n_sides = 6 * plan->n_elements;
  hipLaunchKernelGGL(bndry_interp_kernel, dim3(std::min(n_sides, 8192)), dim3(64), 0, plan->stream, dev, plan->d_bndry,
                     (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, fh.d_side_bndry_stride,
                     plan->d_face_ops, n_sides);
  HIP_CHECK(hipGetLastError());
  if (tmp) {
    HIP_CHECK(hipStreamSynchronize(plan->stream));
    HIP_CHECK(hipFree(tmp));
  }
}

// brick geometry on the mortars (d4est_mesh_compute_mortar_quadrature_quantities on the brick map): constant per mortar;
// the mortar is the face of a cell of the MORTAR's size (half the element on the big side of a hanging face)
__device__ inline void brick_fill_mortar(int f, double hx, double hy, double hz, size_t S, int off, int TT, int T,
                                         double* sj, double* nrm, double* drst_m, double* drst_p, double* hm, double* hp) {
  const int dir = f >> 1;
  const double h[3] = {hx, hy, hz};
  const double J = hx * hy * hz;
  const double sjv = J * (1. / h[dir]);
  for (int k = threadIdx.x; k < T; k += blockDim.x) {
    sj[S + off + k] = sjv;
    hm[S + off + k] = J / sjv;
    hp[S + off + k] = J / sjv;
    for (int d = 0; d < 3; ++d) nrm[3 * S + (size_t)d * TT + off + k] = (d == dir) ? ((f & 1) ? 1. : -1.) : 0.;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double v = (i == j) ? 1. / h[i] : 0.;
        drst_m[9 * S + (size_t)(i + 3 * j) * TT + off + k] = v;
        drst_p[9 * S + (size_t)(i + 3 * j) * TT + off + k] = v;
      }
  }
}

__global__ __launch_bounds__(64) void brick_mortar_kernel(const SideDesc* __restrict__ sd, int n_sides, const int* __restrict__ elem_dq,
                                                          double root_len, double ex, double ey, double ez, double* sj, double* nrm,
                                                          double* drst_m, double* drst_p, double* hm, double* hp) {
  for (int s = blockIdx.x; s < n_sides; s += gridDim.x) {
    const SideDesc d = sd[s];
    if (d.kind == 3) continue;   // hanging side: written by the record kernel
    const double half = (double)elem_dq[s / 6] / root_len / 2.;
    const int T = d.NQ * d.NQ;
    brick_fill_mortar(s % 6, ex * half, ey * half, ez * half, (size_t)d.geom, 0, T, T, sj, nrm, drst_m, drst_p, hm, hp);
  }
}

__global__ __launch_bounds__(64) void brick_mortar_hp_kernel(const HpMortar* __restrict__ md, const HpGeomSrc* __restrict__ gs, int n_rec,
                                                             const int* __restrict__ elem_dq, double root_len, double ex, double ey,
                                                             double ez, double* sj, double* nrm, double* drst_m, double* drst_p,
                                                             double* hm, double* hp) {
  for (int r = blockIdx.x; r < n_rec; r += gridDim.x) {
    const HpMortar m = md[r];
    const HpGeomSrc g = gs[r];
    const double half = (double)elem_dq[m.elem] / root_len / 2. * (m.fm < 1.0 ? 0.5 : 1.0);   // big side: half-size mortar
    brick_fill_mortar(m.face, ex * half, ey * half, ez * half, (size_t)g.S, g.off, g.Ttot, m.NQ * m.NQ, sj, nrm, drst_m, drst_p, hm, hp);
  }
}

void faces_set_geometry_brick(d4est_hip_plan* plan, const int* d_elem_dq, double root_len, const double* extents) {
  FaceHost& fh = g_face_host[plan];
  const size_t T = std::max<size_t>((size_t)plan->total_mortar_nodes, 1);
  double* a[6];
  const size_t mult[6] = {1, 3, 9, 9, 1, 1};
  for (int i = 0; i < 6; ++i) {
    HIP_CHECK(hipMalloc(&a[i], mult[i] * T * sizeof(double)));
    HIP_CHECK(hipMemsetAsync(a[i], 0, mult[i] * T * sizeof(double), plan->stream));
  }
  const double ex = extents[1] - extents[0], ey = extents[3] - extents[2], ez = extents[5] - extents[4];
  const int n_sides = 6 * plan->n_elements;
  if (fh.hp) {
    if (fh.n_rec > 0)
      hipLaunchKernelGGL(brick_mortar_hp_kernel, dim3(std::min(fh.n_rec, 8192)), dim3(64), 0, plan->stream, fh.d_rec, fh.d_gsrc, fh.n_rec,
                         d_elem_dq, root_len, ex, ey, ez, a[0], a[1], a[2], a[3], a[4], a[5]);
  } else if (n_sides > 0) {
    hipLaunchKernelGGL(brick_mortar_kernel, dim3(std::min(n_sides, 8192)), dim3(64), 0, plan->stream, (const SideDesc*)plan->d_side_desc,
                       n_sides, d_elem_dq, root_len, ex, ey, ez, a[0], a[1], a[2], a[3], a[4], a[5]);
  }
  HIP_CHECK(hipGetLastError());
  faces_set_geometry(plan, a[0], a[1], a[2], a[3], a[4], a[5], /*on_device=*/1);
  for (int i = 0; i < 6; ++i) HIP_CHECK(hipFree(a[i]));
}

// ---------------------------------------------------------------------------
// Mortar factors of an analytic tree map on the device (d4est_mesh_compute_mortar_quadrature_quantities with
// DX_compute_method = GEOM_COMPUTE_ANALYTIC, src/Mesh/d4est_mesh.c:858-1108, src/Mesh/d4est_mortars.c:19-190): one unit per
// conforming side / per mortar record.  The (-) cell gives sj, n (COMPUTE_NORMAL_USING_JACOBIAN), dr/dx and hm = J/sj at the
// mortar's quadrature nodes in (-) order; the (+) cell -- evaluated in ITS tree, at ITS nodes -- gives drst_dxyz_p_porder in the
// (+) side's own node and sub-face order and, re-oriented into (-) order, hp.  On the big side of a hanging face the cells are
// the half-size virtual children of the element (d4est_mortars_compute_qcoords_on_mortar, :419-468).  The arrays are written in
// the reference's layout and handed to the same pre-combination as host-supplied factors.
// ---------------------------------------------------------------------------
struct MortarUnit {
  int S, off, off_p, Ttot, NQ, code, boundary, pad;
  CellDesc m, p;
};

__device__ inline void face_ref_point(int f, double ta, double tb, double r[3]) {
  const int dir = f >> 1;
  r[dir] = (f & 1) ? 1.0 : -1.0;
  r[dir == 0 ? 1 : 0] = ta;
  r[dir == 2 ? 1 : 2] = tb;
}

__global__ __launch_bounds__(64) void analytic_mortar_kernel(const MortarUnit* __restrict__ units, int n_units, TreeMapParams P,
                                                             double root_len, const double* __restrict__ quad_nodes /* [NQ][24] */,
                                                             double* sj, double* nrm, double* drst_m, double* drst_p, double* hm, double* hp) {
  for (int ui = blockIdx.x; ui < n_units; ui += gridDim.x) {
    const MortarUnit un = units[ui];
    const int NQ = un.NQ, T = NQ * NQ;
    const double* t = quad_nodes + 24 * NQ;
    const size_t S = (size_t)un.S;
    for (int k = threadIdx.x; k < T; k += blockDim.x) {
      const int a = k % NQ, b = k / NQ;
      double r[3], dxdr[3][3], inv[3][3];
      face_ref_point(un.m.face, t[a], t[b], r);
      cell_dxdr(P, un.m, root_len, r, dxdr);
      const double J = invert3(dxdr, inv);
      const int dir = un.m.face >> 1;
      const double sgn = (un.m.face & 1) ? 1.0 : -1.0;
      double v[3], s2 = 0.0;
      for (int d = 0; d < 3; ++d) { v[d] = sgn * J * inv[dir][d]; s2 += v[d] * v[d]; }
      const double sjv = sqrt(s2);
      sj[S + un.off + k] = sjv;
      hm[S + un.off + k] = J / sjv;
      for (int d = 0; d < 3; ++d) nrm[3 * S + (size_t)d * un.Ttot + un.off + k] = v[d] / sjv;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) drst_m[9 * S + (size_t)(i + 3 * j) * un.Ttot + un.off + k] = inv[i][j];
      if (un.boundary) {
        hp[S + un.off + k] = J / sjv;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) drst_p[9 * S + (size_t)(i + 3 * j) * un.Ttot + un.off_p + k] = inv[i][j];
        continue;
      }
      // (+) side, its own node k
      face_ref_point(un.p.face, t[a], t[b], r);
      cell_dxdr(P, un.p, root_len, r, dxdr);
      (void)invert3(dxdr, inv);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) drst_p[9 * S + (size_t)(i + 3 * j) * un.Ttot + un.off_p + k] = inv[i][j];
      // hp at (-) node k = J/sj of the (+) side at its node reorder(k)
      const int kp = reorder_index(un.code, NQ - 1, a, b);
      face_ref_point(un.p.face, t[kp % NQ], t[kp / NQ], r);
      cell_dxdr(P, un.p, root_len, r, dxdr);
      const double Jp = invert3(dxdr, inv);
      const int dirp = un.p.face >> 1;
      double sp = 0.0;
      for (int d = 0; d < 3; ++d) sp += (Jp * inv[dirp][d]) * (Jp * inv[dirp][d]);
      hp[S + un.off + k] = Jp / sqrt(sp);
    }
  }
}

static CellDesc face_child(const CellDesc& c, int face, int child) {
  // half-size virtual child `child` (z-order on the face) of cell c that touches face `face`
  CellDesc r = c;
  const int dir = face >> 1, h = c.dq / 2;
  const int a0 = dir == 0 ? 1 : 0, a1 = dir == 2 ? 1 : 2;
  r.dq = h;
  r.q[a0] += (child & 1) * h;
  r.q[a1] += (child >> 1) * h;
  if (face & 1) r.q[dir] += h;
  r.face = face;
  return r;
}

void faces_set_geometry_analytic(d4est_hip_plan* plan, const TreeMapParams& P, const std::vector<CellDesc>& elem,
                                 const std::vector<CellDesc>& ghost, double root_len) {
  FaceHost& fh = g_face_host[plan];
  const int ne = plan->n_elements;
  auto cell_of = [&](int ref, int face) {
    if (ref == -1) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: boundary reference");
    CellDesc c;
    if (ref >= 0) c = elem[ref];
    else {
      const int g = -(ref + 2);
      if (g >= (int)ghost.size()) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: ghost element %d has no cell description", g);
      c = ghost[g];
    }
    c.face = face;
    return c;
  };
  std::vector<MortarUnit> units;
  if (fh.hp) {
    std::vector<HpMortar> rec((size_t)fh.n_rec);
    std::vector<HpGeomSrc> gs((size_t)fh.n_rec);
    HIP_CHECK(hipMemcpy(rec.data(), fh.d_rec, rec.size() * sizeof(HpMortar), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(gs.data(), fh.d_gsrc, gs.size() * sizeof(HpGeomSrc), hipMemcpyDeviceToHost));
    for (int e = 0; e < ne; ++e)
      for (int f = 0; f < 6; ++f) {
        const size_t s = 6 * (size_t)e + f;
        const int hang = plan->side_hang[s], nbr = plan->side_nbr[s], f_p = plan->side_nbr_face[s], o = plan->side_orientation[s];
        const int r0 = plan->side_first_rec[s], r1 = plan->side_first_rec[s + 1];
        for (int r = r0; r < r1; ++r) {
          MortarUnit u{};
          u.S = gs[r].S; u.off = gs[r].off; u.off_p = gs[r].off_p; u.Ttot = gs[r].Ttot; u.NQ = rec[r].NQ; u.code = rec[r].code;
          u.boundary = (rec[r].kind == 0);
          if (hang == 0) {
            u.m = cell_of(e, f);
            u.p = u.boundary ? u.m : cell_of(nbr, f_p);
          } else if (hang == 1) {
            const int i = r - r0;
            u.m = face_child(cell_of(e, f), f, i);
            u.p = cell_of(plan->side_nbr4[4 * s + i], f_p);
          } else {
            const int c = plan->side_sub[s];
            u.m = cell_of(e, f);
            u.p = face_child(cell_of(nbr, f_p), f_p, reorient_face_order(f, f_p, o, c));
          }
          units.push_back(u);
        }
      }
  } else {
    std::vector<SideDesc> sd(6 * (size_t)ne);
    if (!sd.empty()) HIP_CHECK(hipMemcpy(sd.data(), plan->d_side_desc, sd.size() * sizeof(SideDesc), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < sd.size(); ++s) {
      MortarUnit u{};
      const int T = sd[s].NQ * sd[s].NQ;
      u.S = sd[s].geom; u.off = 0; u.off_p = 0; u.Ttot = T; u.NQ = sd[s].NQ; u.code = sd[s].code; u.boundary = (sd[s].kind == 0);
      u.m = cell_of((int)(s / 6), (int)(s % 6));
      u.p = u.boundary ? u.m : cell_of(plan->side_nbr[s], plan->side_nbr_face[s]);
      units.push_back(u);
    }
  }
  // quadrature nodes of every mortar degree in use
  std::vector<double> qn((size_t)25 * 24, 0.0);
  for (const MortarUnit& u : units) {
    if (u.NQ > 24) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: mortar degree %d", u.NQ - 1);
    if (qn[(size_t)24 * u.NQ + u.NQ - 1] == 0.0 || u.NQ == 1) {
      std::vector<double> x, w;
      if (plan->quad_type == QUAD_LEGENDRE) Tables1D::gauss(u.NQ - 1, x, w);
      else Tables1D::lobatto(u.NQ - 1, x, w);
      for (int i = 0; i < u.NQ; ++i) qn[(size_t)24 * u.NQ + i] = x[i];
    }
  }
  const size_t T = std::max<size_t>((size_t)plan->total_mortar_nodes, 1);
  double* a[6];
  const size_t mult[6] = {1, 3, 9, 9, 1, 1};
  for (int i = 0; i < 6; ++i) {
    HIP_CHECK(hipMalloc(&a[i], mult[i] * T * sizeof(double)));
    HIP_CHECK(hipMemsetAsync(a[i], 0, mult[i] * T * sizeof(double), plan->stream));
  }
  MortarUnit* d_units = upload_vec(units);
  double* d_qn = upload_vec(qn);
  if (!units.empty())
    hipLaunchKernelGGL(analytic_mortar_kernel, dim3(std::min((int)units.size(), 8192)), dim3(64), 0, plan->stream, d_units, (int)units.size(), P,
                       root_len, d_qn, a[0], a[1], a[2], a[3], a[4], a[5]);
  HIP_CHECK(hipGetLastError());
  faces_set_geometry(plan, a[0], a[1], a[2], a[3], a[4], a[5], /*on_device=*/1);
  for (int i = 0; i < 6; ++i) HIP_CHECK(hipFree(a[i]));
  HIP_CHECK(hipFree(d_units));
  HIP_CHECK(hipFree(d_qn));
}

__global__ __launch_bounds__(256) void robin_setup_kernel(const double* __restrict__ sj, const double* __restrict__ coeff,
                                                          const double* __restrict__ rhs, double* __restrict__ c,
                                                          double* __restrict__ r, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    c[i] = sj[i] * coeff[i];
    r[i] = sj[i] * rhs[i];
  }
}

// Robin boundary data at the mortar quadrature nodes of the boundary sides (indexed like sj: side stride S + k).
// coeff_quad == NULL switches the boundary sides back to Dirichlet.
void faces_set_robin(d4est_hip_plan* plan, const double* coeff_quad, const double* rhs_quad, int on_device) {
  FaceHost& fh = g_face_host[plan];
  if (!plan->has_face_geometry) D4EST_HIP_ABORT("plan_set_robin_values: call d4est_hip_plan_set_mortar_geometry first (needs sj)");
  if (!coeff_quad) {
    fh.robin = false;
    plan->bc_inhomogeneous = fh.dirichlet_nonzero;   // back to Dirichlet: the values set earlier are still in d_bndry
    return;
  }
  if (!rhs_quad) D4EST_HIP_ABORT("plan_set_robin_values: rhs_quad is NULL");
  const size_t tm = (size_t)plan->total_mortar_nodes;
  if (!fh.d_robin_c) {
    HIP_CHECK(hipMalloc(&fh.d_robin_c, std::max<size_t>(tm, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(&fh.d_robin_r, std::max<size_t>(tm, 1) * sizeof(double)));
  }
  fh.robin = true;
  plan->bc_inhomogeneous = true;
  if (tm == 0) return;
  const double* dc = coeff_quad;
  const double* dr = rhs_quad;
  double *tc = nullptr, *tr = nullptr;
  if (!on_device) {
    HIP_CHECK(hipMalloc(&tc, tm * sizeof(double)));
    HIP_CHECK(hipMalloc(&tr, tm * sizeof(double)));
    HIP_CHECK(hipMemcpy(tc, coeff_quad, tm * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(tr, rhs_quad, tm * sizeof(double), hipMemcpyHostToDevice));
    dc = tc;
    dr = tr;
  }
  hipLaunchKernelGGL(robin_setup_kernel, dim3(1024), dim3(256), 0, plan->stream, fh.d_sj, dc, dr, fh.d_robin_c, fh.d_robin_r, tm);
  HIP_CHECK(hipGetLastError());
  if (tc) {
    HIP_CHECK(hipStreamSynchronize(plan->stream));
    HIP_CHECK(hipFree(tc));
    HIP_CHECK(hipFree(tr));
  }
}

// resident workgroups per CU assumed by the persistent fast kernels (experiment knob: D4EST_HIP_FACE_WG_PER_CU)
static int face_wg_per_cu() {
  static int v = -1;
  if (v < 0) {
    const char* e = std::getenv("D4EST_HIP_FACE_WG_PER_CU");
    v = e ? std::atoi(e) : 4;
    if (v < 1) v = 4;
  }
  return v;
}

static void debug_occupancy_once() {
  static bool done = false;
  if (done || !std::getenv("D4EST_HIP_DEBUG_OCC")) return;
  done = true;
  int nt = -1, nf = -1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nt, reinterpret_cast<const void*>(trace_wave_kernel), 384, 0);
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, reinterpret_cast<const void*>(flux_wave_kernel<false, true>), 384, 0);
  hipFuncAttributes at{}, af{};
  (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(trace_wave_kernel));
  (void)hipFuncGetAttributes(&af, reinterpret_cast<const void*>(flux_wave_kernel<false, true>));
  std::fprintf(stderr, "[d4est_hip] occupancy: trace_wave %d wg/CU (regs %d, lds %zu, scratch %zu) flux_wave %d wg/CU (regs %d, lds %zu, scratch %zu)\n",
               nt, at.numRegs, at.sharedSizeBytes, at.localSizeBytes, nf, af.numRegs, af.sharedSizeBytes, af.localSizeBytes);
}

static size_t generic_lds_bytes(const d4est_hip_plan* plan) { return (size_t)plan->max_face_lds_doubles * sizeof(double); }

void launch_traces(d4est_hip_plan* plan, const double* u, double* trace, bool ghost, const int* elist, int n_list, int parts) {
  FaceHost& fh = g_face_host[plan];
  if (ghost) {
    if (fh.hp) D4EST_HIP_ABORT("compute_ghost_traces: plans with hanging faces take their ghost traces from the trace exchange (d4est_hip_plan_*_sub offsets), not from whole ghost elements");
    if (fh.n_ghost_sides == 0) return;
    const size_t lds = generic_lds_bytes(plan);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ghost_trace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ghost_trace_kernel, dim3(std::min(fh.n_ghost_sides, 16384)), dim3(256), lds, plan->stream, u, trace,
                       fh.d_ghost_sides, plan->d_face_ops, fh.n_ghost_sides, fh.fld_stride);
    HIP_CHECK(hipGetLastError());
    return;
  }
  // elist: only these elements' traces are needed (the tiled MFMA kernels take the list; the other families compute every element)
  const bool listed = elist && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 &&
                      ((fh.hp && (fh.hp_split || (fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16))) ||
                       (!fh.hp && ((plan->face_fast && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 1) || (!plan->face_fast && fh.max_N <= 16 && fh.max_NQ <= 16))));
  if (!listed) elist = nullptr;
  const int n = listed ? n_list : plan->n_elements;
  // the two conforming families' lists of this launch: the whole plan's, or the hybrid operator's ring list split the same way
  const bool fam = fh.family_split && (!elist || (elist == fh.hy_ring && fh.d_ring_small));
  const int* fam_small = elist ? fh.d_ring_small : fh.d_fam_small;
  const int* fam_big = elist ? fh.d_ring_big : fh.d_fam_big;
  const int n_fam_small = elist ? fh.n_ring_small : fh.n_fam_small, n_fam_big = elist ? fh.n_ring_big : fh.n_fam_big;
  if (parts != 3 && !(fh.hp && fh.hp_split)) D4EST_HIP_ABORT("launch_traces: parts = %d on a plan without the hp split", parts);
  if (n == 0 && !(fh.hp && fh.hp_split && (parts & 2))) return;
  if (fh.hp && fh.hp_split) {
    // hp split: every conforming side from the fast conforming kernel, then the hanging sides of the elements that have one (the
    // record kernel overwrites the blocks the first kernel filled for those sides)
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    if (n > 0 && (parts & 1) && fh.hp_split_fast) {
      const int resident = 8 * cus;
      const int rounds = (n + resident - 1) / resident;
      const int grid = (n + rounds - 1) / rounds;
      hipLaunchKernelGGL(trace_mfma_kernel, dim3(grid), dim3(192), 0, plan->stream, u, trace, (const SideDesc*)plan->d_side_desc,
                         (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, n, elist);
    } else if (n > 0 && (parts & 1)) {
      // degrees above 7: the conforming sides through the tiled kernels -- both conforming families on their lists where the plan has them
      const int max_local_n = fh.max_local_N;
      const size_t lds = (size_t)(max_local_n * 272 + 3 * 2 * 16 * 34) * sizeof(double);
      if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_mfma16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
      if (fam) {
        if (n_fam_small > 0) {
          const int ns_ = n_fam_small, resident = 8 * cus, rounds = (ns_ + resident - 1) / resident, grid = (ns_ + rounds - 1) / rounds;
          hipLaunchKernelGGL(trace_mfma_kernel, dim3(grid), dim3(192), 0, plan->stream, u, trace, (const SideDesc*)plan->d_side_desc,
                             (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, ns_, fam_small);
        }
        if (n_fam_big > 0)
          hipLaunchKernelGGL(trace_mfma16_kernel, dim3(std::min(n_fam_big, per_cu * cus)), dim3(192), lds, plan->stream, u, trace,
                             (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, n_fam_big, max_local_n,
                             fam_big);
      } else {
        hipLaunchKernelGGL(trace_mfma16_kernel, dim3(std::min(n, per_cu * cus)), dim3(192), lds, plan->stream, u, trace,
                           (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, n, max_local_n, elist);
      }
    }
    if (fh.n_units > 0 && (parts & 2)) {
      const int uj = fh.hp_max_N | 1, uk = (fh.hp_max_N * uj) | 1;
      const size_t lds = (size_t)(((fh.hp_max_N * uk + 1) & ~1) + 4 * 16 * 34) * sizeof(double);
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
      hipLaunchKernelGGL(trace_unit_kernel, dim3(std::min(fh.n_units, per_cu * cus)), dim3(256), lds, plan->stream, u, trace, fh.d_rec, fh.d_units,
                         (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, fh.n_units, fh.hp_max_N);
    } else if (fh.n_hang_elems > 0 && (parts & 2)) {
      const size_t lds = (size_t)(fh.hp_max_N * 272 + 3 * 2 * 16 * 34) * sizeof(double);
      if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_hp_mfma16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(trace_hp_mfma16_kernel, dim3(std::min(fh.n_hang_elems, 4 * cus)), dim3(192), lds, plan->stream, u, trace, fh.d_rec,
                         fh.d_side_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, fh.n_hang_elems, fh.hp_max_N,
                         (const int*)fh.d_hang_elems, 1);
    }
  } else if (fh.hp && fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    const size_t lds = (size_t)(fh.hp_max_N * 272 + 3 * 2 * 16 * 34) * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_hp_mfma16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(trace_hp_mfma16_kernel, dim3(std::min(n, 4 * (plan->n_cus > 0 ? plan->n_cus : 256))), dim3(192), lds, plan->stream, u, trace,
                       fh.d_rec, fh.d_side_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, n, fh.hp_max_N, elist, 0);
  } else if (fh.hp) {
    const size_t lds = fh.hp_lds_doubles * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_hp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(trace_hp_kernel, dim3(std::min(n, 16384)), dim3(256), lds, plan->stream, u, trace, fh.d_rec, fh.d_elem_first,
                       (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, n, fh.hp_fld_stride);
  } else if (plan->face_fast && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    const bool mfma = plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 1;
    // resident workgroups per CU: 8 of the 3-wave MFMA kernel (47 VGPRs, 11.5 KB LDS), 4 of the 6-wave kernel
    static const bool wg_override = std::getenv("D4EST_HIP_FACE_WG_PER_CU") != nullptr;   // environment knobs are read once
    const int resident = (wg_override ? face_wg_per_cu() : (mfma ? 8 : 4)) * cus;
    const int rounds = (n + resident - 1) / resident;
    const int grid = (n + rounds - 1) / rounds;
    debug_occupancy_once();
    if (mfma)   // auto: MFMA form of the two interpolation passes
      hipLaunchKernelGGL(trace_mfma_kernel, dim3(grid), dim3(192), 0, plan->stream, u, trace, (const SideDesc*)plan->d_side_desc,
                         (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, n, elist);
    else
      hipLaunchKernelGGL(trace_wave_kernel, dim3(grid), dim3(384), 0, plan->stream, u, trace, (const SideDesc*)plan->d_side_desc,
                         (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, n, fh.uni);
  } else if (fam && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    if (n_fam_small > 0) {
      const int ns_ = n_fam_small, resident = 8 * cus, rounds = (ns_ + resident - 1) / resident, grid = (ns_ + rounds - 1) / rounds;
      hipLaunchKernelGGL(trace_mfma_kernel, dim3(grid), dim3(192), 0, plan->stream, u, trace, (const SideDesc*)plan->d_side_desc,
                         (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, ns_, fam_small);
    }
    const int max_local_n = fh.max_local_N;
    const size_t lds = (size_t)(max_local_n * 272 + 3 * 2 * 16 * 34) * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_mfma16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
    if (n_fam_big > 0)
      hipLaunchKernelGGL(trace_mfma16_kernel, dim3(std::min(n_fam_big, per_cu * cus)), dim3(192), lds, plan->stream, u, trace,
                         (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, n_fam_big, max_local_n,
                         fam_big);
  } else if (fh.max_N <= 16 && fh.max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    // p = 8 .. 15: tiled MFMA trace kernel (descriptors with unpadded N x N derivative matrices)
    // only local elements are copied to LDS: size the copy of u by the largest LOCAL degree
    const int max_local_n = fh.max_local_N;   // (cached at set-up: no per-launch walk over the elements)
    const size_t lds = (size_t)(max_local_n * 272 + 3 * 2 * 16 * 34) * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_mfma16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
    hipLaunchKernelGGL(trace_mfma16_kernel, dim3(std::min(n, per_cu * (plan->n_cus > 0 ? plan->n_cus : 256))), dim3(192), lds, plan->stream, u, trace,
                       (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, n, max_local_n, elist);
  } else {
    const size_t lds = generic_lds_bytes(plan);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(trace_generic_kernel, dim3(std::min(n, 16384)), dim3(256), lds, plan->stream, u, trace,
                       (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, n,
                       fh.fld_stride);
  }
  HIP_CHECK(hipGetLastError());
}

// true when launch_flux runs the kernel that can carry the Chebyshev update in its epilogue
bool flux_can_fuse_update(d4est_hip_plan* plan) {
  FaceHost& fh = g_face_host[plan];
  return plan->has_faces && plan->n_elements > 0 && !fh.hp && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 && !hybrid_active(plan) &&
         (plan->face_fast || (fh.max_N <= 16 && fh.max_NQ <= 16));
}

void launch_flux(d4est_hip_plan* plan, const double* trace, const double* ghost_trace, double* Au, const ChebyFuse* cf, const int* elist,
                 int n_list, int parts) {
  FaceHost& fh = g_face_host[plan];
  if (cf && !(flux_can_fuse_update(plan) || (elist && hybrid_active(plan) && !fh.hp))) D4EST_HIP_ABORT("launch_flux: fused update requested on a plan whose flux kernel cannot carry it");
  if (!plan->has_faces || !plan->has_face_geometry) D4EST_HIP_ABORT("apply flux: plan_set_faces / plan_set_mortar_geometry were not called");
  if (plan->n_elements == 0) return;
  if (fh.n_ghost_sides > 0 && !ghost_trace) D4EST_HIP_ABORT("apply flux: plan has %d ghost sides but no ghost trace buffer was given", fh.n_ghost_sides);
  const int n = elist ? n_list : plan->n_elements;
  if (parts != 3 && !(fh.hp && fh.hp_split)) D4EST_HIP_ABORT("launch_flux: parts = %d on a plan without the hp split", parts);
  if (n == 0 && !(fh.hp && fh.hp_split && (parts & 2))) return;
  // (a fused update on a list: the hybrid operator's dirty elements only -- cf->u_out set, see apply_operator)
  if (elist && cf && !(elist == fh.hy_dirty_any && cf->u_out)) D4EST_HIP_ABORT("launch_flux: a fused update cannot ride on this element list");
  // the two conforming families' lists of this launch: the whole plan's, or the hybrid operator's dirty list split the same way
  const bool fam = fh.family_split && (!elist || (elist == fh.hy_dirty && fh.d_dirty_small));
  const int* fam_small = elist ? fh.d_dirty_small : fh.d_fam_small;
  const int* fam_big = elist ? fh.d_dirty_big : fh.d_fam_big;
  const int n_fam_small = elist ? fh.n_dirty_small : fh.n_fam_small, n_fam_big = elist ? fh.n_dirty_big : fh.n_fam_big;
  if (fh.hp && fh.hp_split) {
    // hp split (see launch_traces): the conforming sides' terms from the fast kernel, then the hanging sides' from the record kernel
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    if (n > 0 && (parts & 1) && fh.hp_split_fast) {
      const int resident = face_wg_per_cu() * cus;
      const int rounds = (n + resident - 1) / resident;
      const int grid = (n + rounds - 1) / rounds;
      static const bool no_remap_s = std::getenv("D4EST_HIP_NO_XCD_REMAP") != nullptr;
      const int chunk_s = (!elist && n % 8 == 0 && grid % 8 == 0 && !no_remap_s) ? n / 8 : 0;   // XCD-aware element order, as on conforming plans
      hipLaunchKernelGGL((flux_wave_kernel<false, true>), dim3(grid), dim3(384), 0, plan->stream, trace, ghost_trace, Au,
                         (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, plan->d_face_geom,
                         plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, chunk_s, ChebyFuse{}, elist);
    } else if (n > 0 && (parts & 1)) {
      if (fam) {
        if (n_fam_small > 0) {
          const int ns_ = n_fam_small, resident = face_wg_per_cu() * cus, rounds = (ns_ + resident - 1) / resident, grid = (ns_ + rounds - 1) / rounds;
          hipLaunchKernelGGL((flux_wave_kernel<false, true>), dim3(grid), dim3(384), 0, plan->stream, trace, ghost_trace, Au,
                             (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, plan->d_face_geom, plan->d_bndry,
                             fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, ns_, 0, ChebyFuse{}, fam_small);
        }
        if (n_fam_big > 0)
          hipLaunchKernelGGL(flux_mfma16_kernel<false>, dim3(std::min(n_fam_big, 8 * cus)), dim3(192), 0, plan->stream, trace, ghost_trace, Au,
                             (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, plan->d_face_geom, plan->d_bndry,
                             fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n_fam_big, 0, ChebyFuse{}, fam_big);
      } else {
        hipLaunchKernelGGL(flux_mfma16_kernel<false>, dim3(std::min(n, 8 * cus)), dim3(192), 0, plan->stream, trace, ghost_trace, Au,
                           (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, plan->d_face_geom, plan->d_bndry,
                           fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, 0, ChebyFuse{}, elist);
      }
    }
    if (fh.n_units > 0 && (parts & 2))
    {
      auto go = [&](auto kern, int per_cu) {
        hipLaunchKernelGGL(kern, dim3(std::min(fh.n_hang_elems, per_cu * cus)), dim3(256), 0, plan->stream, trace, ghost_trace, Au, fh.d_rec,
                           fh.d_units, fh.d_unit_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, plan->d_face_geom,
                           plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, fh.n_hang_elems, ChebyFuse{});
      };
      if (fh.hp_max_N <= 8) go(flux_unit_kernel<false, 8>, 8);
      else go(flux_unit_kernel<false, 16>, 2);
    }
    else if (fh.n_hang_elems > 0 && (parts & 2))
      hipLaunchKernelGGL(flux_hp_mfma16_kernel, dim3(std::min(fh.n_hang_elems, 8 * cus)), dim3(192), 0, plan->stream, trace, ghost_trace, Au,
                         fh.d_rec, fh.d_side_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, plan->d_face_geom,
                         plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, fh.n_hang_elems,
                         (const int*)fh.d_hang_elems, 1);
  } else if (elist && ((fh.hp && !(fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0)) ||
                       (!fh.hp && !(plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 && (plan->face_fast || (fh.max_N <= 16 && fh.max_NQ <= 16)))))) {
    D4EST_HIP_ABORT("launch_flux: no list form of this plan's flux kernels (the hybrid operator is not set up for such plans)");
  } else if (fh.hp && fh.hp_max_N <= 16 && fh.hp_max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    hipLaunchKernelGGL(flux_hp_mfma16_kernel, dim3(std::min(n, 8 * (plan->n_cus > 0 ? plan->n_cus : 256))), dim3(192), 0, plan->stream, trace, ghost_trace, Au,
                       fh.d_rec, fh.d_side_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, plan->d_face_geom,
                       plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, elist, 0);
  } else if (fh.hp) {
    const size_t lds = fh.hp_lds_doubles * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(flux_hp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(flux_hp_kernel, dim3(std::min(n, 16384)), dim3(256), lds, plan->stream, trace, ghost_trace, Au, fh.d_rec, fh.d_elem_first,
                       (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, plan->d_face_geom, plan->d_bndry,
                       fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, fh.hp_fld_stride);
  } else if (plan->face_fast && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    // persistent grid: 3 workgroups per CU are resident (LDS), each loops over elements
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    const int resident = face_wg_per_cu() * cus;
    const int rounds = (n + resident - 1) / resident;
    const int grid = (n + rounds - 1) / rounds;
    static const bool no_remap = std::getenv("D4EST_HIP_NO_XCD_REMAP") != nullptr;
    const int chunk = (!elist && n % 8 == 0 && grid % 8 == 0 && !no_remap) ? n / 8 : 0;
    const bool subdomain_plan = plan->tuning[D4EST_HIP_TUNE_GHOST_ALIAS] > 0;   // see flux_wave_kernel: INPLACE
#define D4EST_HIP_LAUNCH_FLUX_WAVE(FUSE_, INPLACE_, CF_)                                                                                 \
  hipLaunchKernelGGL((flux_wave_kernel<FUSE_, INPLACE_>), dim3(grid), dim3(384), 0, plan->stream, trace, ghost_trace, Au,               \
                     (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, plan->d_face_geom,        \
                     plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, chunk, CF_, elist)
    if (cf && subdomain_plan) D4EST_HIP_LAUNCH_FLUX_WAVE(true, false, *cf);
    else if (cf) D4EST_HIP_LAUNCH_FLUX_WAVE(true, true, *cf);
    else if (subdomain_plan) D4EST_HIP_LAUNCH_FLUX_WAVE(false, false, ChebyFuse{});
    else D4EST_HIP_LAUNCH_FLUX_WAVE(false, true, ChebyFuse{});
#undef D4EST_HIP_LAUNCH_FLUX_WAVE
  } else if (fam && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    // (cf: the Chebyshev update in both families' epilogues -- every element is on exactly one of the two lists)
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    const ChebyFuse cfv = cf ? *cf : ChebyFuse{};
    if (n_fam_small > 0) {
      const int ns_ = n_fam_small, resident = face_wg_per_cu() * cus, rounds = (ns_ + resident - 1) / resident, grid = (ns_ + rounds - 1) / rounds;
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(384), 0, plan->stream, trace, ghost_trace, Au,
                           (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops, plan->d_face_geom, plan->d_bndry,
                           fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, ns_, 0, cfv, fam_small);
      };
      if (cf) go(flux_wave_kernel<true, true>); else go(flux_wave_kernel<false, true>);
    }
    if (n_fam_big > 0) {
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(std::min(n_fam_big, 8 * cus)), dim3(192), 0, plan->stream, trace, ghost_trace, Au,
                           (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, plan->d_face_geom, plan->d_bndry,
                           fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n_fam_big, 0, cfv, fam_big);
      };
      if (cf) go(flux_mfma16_kernel<true>); else go(flux_mfma16_kernel<false>);
    }
  } else if (fh.max_N <= 16 && fh.max_NQ <= 16 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    const int grid16 = std::min(n, 8 * (plan->n_cus > 0 ? plan->n_cus : 256));
    static const bool no_remap16 = std::getenv("D4EST_HIP_NO_XCD_REMAP") != nullptr;
    const int chunk16 = (!elist && n % 8 == 0 && grid16 % 8 == 0 && !no_remap16) ? n / 8 : 0;
    if (cf)
      hipLaunchKernelGGL(flux_mfma16_kernel<true>, dim3(grid16), dim3(192), 0, plan->stream, trace,
                         ghost_trace, Au, (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops,
                         plan->d_face_geom, plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, chunk16, *cf, elist);
    else
      hipLaunchKernelGGL(flux_mfma16_kernel<false>, dim3(grid16), dim3(192), 0, plan->stream, trace,
                         ghost_trace, Au, (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops,
                         plan->d_face_geom, plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n, chunk16,
                         ChebyFuse{}, elist);
  } else {
    const size_t lds = generic_lds_bytes(plan);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(flux_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(flux_generic_kernel, dim3(std::min(n, 16384)), dim3(256), lds, plan->stream, trace, ghost_trace, Au,
                       (const SideDesc*)plan->d_side_desc, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops,
                       plan->d_face_geom, plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, n,
                       fh.fld_stride);
  }
  HIP_CHECK(hipGetLastError());
}

void launch_flux_direct(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf, int vol_term) {
  FaceHost& fh = g_face_host[plan];
  launch_direct_faces(plan, u, ghost_trace, Au, cf, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, vol_term);
}

void launch_flux_hybrid_clean(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, int phase, const DirectFuse* cf) {
  FaceHost& fh = g_face_host[plan];
  launch_hybrid_clean(plan, u, ghost_trace, Au, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, phase, cf);
}

void launch_flux_units(d4est_hip_plan* plan, const double* trace, const double* ghost_trace, double* Au, const ChebyFuse* cf) {
  FaceHost& fh = g_face_host[plan];
  if (!(fh.hp && fh.hp_split)) D4EST_HIP_ABORT("launch_flux_units: the plan has no hp split");
  if (!cf) { launch_flux(plan, trace, ghost_trace, Au, nullptr, nullptr, 0, 2); return; }
  if (fh.n_hang_elems == 0) return;
  if (fh.n_units == 0) D4EST_HIP_ABORT("launch_flux_units: a fused update needs the unit form of the record kernels");
  if (!cf->u_out || cf->u_out == cf->u) D4EST_HIP_ABORT("launch_flux_units: the fused update needs a second vector");
  const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
  auto go = [&](auto kern, int per_cu) {
    hipLaunchKernelGGL(kern, dim3(std::min(fh.n_hang_elems, per_cu * cus)), dim3(256), 0, plan->stream, trace, ghost_trace, Au, fh.d_rec,
                       fh.d_units, fh.d_unit_first, (const ElemDesc*)fh.d_elem_desc_generic, plan->d_face_ops, fh.d_hp_ops, plan->d_face_geom,
                       plan->d_bndry, fh.robin ? fh.d_robin_c : nullptr, fh.robin ? fh.d_robin_r : nullptr, fh.n_hang_elems, *cf);
  };
  if (fh.hp_max_N <= 8) go(flux_unit_kernel<true, 8>, 8);
  else go(flux_unit_kernel<true, 16>, 2);
  HIP_CHECK(hipGetLastError());
}
bool faces_hp(d4est_hip_plan* plan) { return g_face_host[plan].hp; }
bool faces_hp_split(d4est_hip_plan* plan) {
  FaceHost& fh = g_face_host[plan];
  return fh.hp && fh.hp_split;
}
bool faces_have_units(d4est_hip_plan* plan) {
  FaceHost& fh = g_face_host[plan];
  return fh.hp && fh.hp_split && (fh.n_hang_elems == 0 || fh.n_units > 0);
}

void faces_destroy(d4est_hip_plan* plan) {
  direct_destroy(plan);
  hybrid_destroy(plan);
  auto it = g_face_host.find(plan);
  if (it != g_face_host.end()) {
    FaceHost& fh = it->second;
    (void)hipFree(fh.d_side_deg_m); (void)hipFree(fh.d_side_deg_p); (void)hipFree(fh.d_side_bndry_stride);
    (void)hipFree(fh.d_ghost_sides); (void)hipFree(fh.d_elem_desc_generic);
    (void)hipFree(fh.d_sj); (void)hipFree(fh.d_robin_c); (void)hipFree(fh.d_robin_r);
    (void)hipFree(fh.d_fam_small); (void)hipFree(fh.d_fam_big);
    (void)hipFree(fh.d_rec); (void)hipFree(fh.d_gsrc); (void)hipFree(fh.d_elem_first); (void)hipFree(fh.d_side_first); (void)hipFree(fh.d_hp_ops); (void)hipFree(fh.d_hang_elems);
    (void)hipFree(fh.d_units); (void)hipFree(fh.d_unit_first);
    (void)hipFree(fh.d_ring_small); (void)hipFree(fh.d_ring_big); (void)hipFree(fh.d_dirty_small); (void)hipFree(fh.d_dirty_big);
    g_face_host.erase(it);
  }
  (void)hipFree(plan->d_elem_desc);
  (void)hipFree(plan->d_side_desc); (void)hipFree(plan->d_face_ops);
  (void)hipFree(plan->d_face_geom); (void)hipFree(plan->d_bndry); (void)hipFree(plan->d_trace);
}

}  // namespace d4est_hip
