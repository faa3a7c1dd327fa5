// Face (mortar) part of the weak Laplacian: SIPG flux on conforming mortars and Dirichlet boundaries.
//
// Replaces, for all local (element, face) sides at once,
//   d4est_laplacian_compute_dudr              (src/dGMath/d4est_laplacian.c:237-282)  -> only the face TRACES are formed
//   d4est_laplacian_flux_interface/_boundary  (src/dGMath/d4est_laplacian_flux.c:232-1014, :23-230)
//   d4est_laplacian_flux_sipg_interface/_dirichlet (src/dGMath/d4est_laplacian_flux_sipg.c:494-942, :15-336)
//   d4est_mortars_compute_flux_on_local_elements   (src/Mesh/d4est_mortars.c:601-840; the serial p4est_iterate walk)
//
// Design: element-centric and two-phase.
//  (1) trace kernel: every element writes the traces of u and of du/dr_{0,1,2} on its six faces
//      (4 N^2 doubles per face) -- the only data a neighbour (or another GPU) ever needs, instead of the
//      reference's three full dudr vectors and whole-element ghost copies.
//  (2) flux kernel: one workgroup per element walks its six sides, reads its own and the neighbour's trace,
//      evaluates the three SIPG terms at the mortar quadrature nodes, integrates, projects back, lifts and
//      applies D^T into an LDS accumulator, and adds it to Au_e once: race-free and deterministic, no atomics.
// The reference's 24 doubles of mortar geometry per quadrature node are pre-combined at set-up into 7:
//   am_i = sum_d sj n_d (dr_i/dx_d)^-,  ap_i = sum_d sj n_d (dr_i/dx_d)^+ (re-ordered to the (-) side),  s3 = sj sigma.
#include <algorithm>
#include <map>
#include <tuple>

#include "d4est_hip_internal.h"
#include "d4est_hip_tables.h"

namespace d4est_hip {

struct SideDesc {
  int kind;           // 0 boundary, 1 interface with local (+), 2 interface with ghost (+)
  int f_p;            // face of the (+) element
  int code;           // flip0 | flip1<<1 | transpose<<2   (dGMath/d4est_operators.c:2031-2081)
  int Np;             // nodes/dir of the (+) element
  int NQ;             // mortar quadrature nodes/dir
  int offC_m, offC_p; // offsets into face_ops: (NQ x N) and (NQ x Np) side -> mortar-quadrature operators
  int offE;           // (N x NQ) mortar-quadrature -> side operator (P^T I^T W)
  int geom;           // scalar stride S of the side (face_geom at 7*S)
  int bndry;          // offset of Dirichlet values (boundary sides)
  long long nbr_trace;  // offset of the (+) element's trace block (local or ghost buffer)
};

struct ElemDesc {
  int N;                // nodes per direction
  int ns;               // nodal stride
  long long trace_off;  // offset of the element's trace block
  int offD;             // offset of the (N x N) derivative matrix inside face_ops
  int pad;
};

__host__ __device__ inline int face_fix(int f, int N) { return (f & 1) ? (N - 1) : 0; }

// volume index of face node (a,b) of face f (a fastest; tangential axes in increasing order)
__device__ inline int face_vol_index(int f, int N, int a, int b) {
  const int dir = f >> 1, fix = face_fix(f, N);
  if (dir == 0) return fix + N * (a + N * b);
  if (dir == 1) return a + N * (fix + N * b);
  return a + N * (b + N * fix);
}

// ---------------------------------------------------------------------------
// (1) traces: T[e][f][c][a + N b], c = 0: u, c = 1..3: du/dr_{c-1}
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_kernel(const double* __restrict__ u, double* __restrict__ trace,
                                                    const int* __restrict__ elem_N, const int* __restrict__ elem_ns,
                                                    const long long* __restrict__ trace_offset,
                                                    const double* const* __restrict__ elem_D, int n_elem) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const int N = elem_N[e], N2 = N * N, N3 = N2 * N;
    const double* __restrict__ D = elem_D[e];
    double* ue = smem;
    double* Ds = smem + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) ue[i] = u[elem_ns[e] + i];
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = D[i];
    __syncthreads();
    double* T = trace + trace_offset[e];
    for (int idx = threadIdx.x; idx < 24 * N2; idx += blockDim.x) {
      const int f = idx / (4 * N2), c = (idx / N2) & 3, ab = idx % N2, a = ab % N, b = ab / N;
      const int v = face_vol_index(f, N, a, b);
      double val;
      if (c == 0) {
        val = ue[v];
      } else {
        const int d = c - 1;
        const int stride = (d == 0) ? 1 : (d == 1 ? N : N2);
        const int pos = (v / stride) % N;  // index along direction d
        const int base = v - pos * stride;
        val = 0.0;
        for (int i = 0; i < N; ++i) val = fma(Ds[pos * N + i], ue[base + i * stride], val);
      }
      T[idx] = val;
    }
    __syncthreads();
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

__device__ inline int reorder_index(int code, int deg, int a, int b) {
  // out(a,b) = in(a2,b2) for out = transpose?(flip1?(flip0?(in)))  (dGMath/d4est_operators.c:2044-2081)
  int a1 = (code & 4) ? b : a, b1 = (code & 4) ? a : b;
  if (code & 2) b1 = deg - b1;
  if (code & 1) a1 = deg - a1;
  return a1 + (deg + 1) * b1;
}

__global__ __launch_bounds__(256) void face_geom_kernel(const SideDesc* __restrict__ sd, const int* __restrict__ side_deg_m,
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
// (2) flux kernel: one workgroup per element
// ---------------------------------------------------------------------------
// out[r x r'] (rows x rows) = (op (x) op) in[cols x cols], op is rows x cols row-major; tmp holds rows x cols
__device__ inline void apply2d(const double* __restrict__ op, int rows, int cols, const double* in, double* tmp, double* out,
                               int nfields, int in_stride, int out_stride, int tmp_stride) {
  // pass 1: tmp(a', b) = sum_a op[a'][a] in(a, b)
  for (int idx = threadIdx.x; idx < nfields * rows * cols; idx += blockDim.x) {
    const int fld = idx / (rows * cols), r = idx % (rows * cols), ap = r % rows, b = r / rows;
    const double* x = in + fld * in_stride + cols * b;
    double s = 0.0;
    for (int a = 0; a < cols; ++a) s = fma(op[ap * cols + a], x[a], s);
    tmp[fld * tmp_stride + ap + rows * b] = s;
  }
  __syncthreads();
  // pass 2: out(a', b') = sum_b op[b'][b] tmp(a', b)
  for (int idx = threadIdx.x; idx < nfields * rows * rows; idx += blockDim.x) {
    const int fld = idx / (rows * rows), r = idx % (rows * rows), ap = r % rows, bp = r / rows;
    const double* x = tmp + fld * tmp_stride + ap;
    double s = 0.0;
    for (int b = 0; b < cols; ++b) s = fma(op[bp * cols + b], x[rows * b], s);
    out[fld * out_stride + ap + rows * bp] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void flux_kernel(const double* __restrict__ trace, const double* __restrict__ ghost_trace,
                                                   double* __restrict__ Au, const SideDesc* __restrict__ sd,
                                                   const int* __restrict__ elem_N, const int* __restrict__ elem_ns,
                                                   const long long* __restrict__ trace_offset,
                                                   const double* const* __restrict__ elem_D,
                                                   const double* __restrict__ face_ops, const double* __restrict__ geom,
                                                   const double* __restrict__ bndry, int n_elem, int fld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // LDS carve: fields A[8][fld_stride], B[8][fld_stride], tmp[8][fld_stride], acc[N^3], D[N^2]
  double* A = smem;
  double* Bq = A + 8 * fld_stride;
  double* tmp = Bq + 8 * fld_stride;
  double* acc = tmp + 8 * fld_stride;
  for (int e = blockIdx.x; e < n_elem; e += gridDim.x) {
    const int N = elem_N[e], N2 = N * N, N3 = N2 * N;
    double* Ds = acc + N3;
    for (int i = threadIdx.x; i < N3; i += blockDim.x) acc[i] = 0.0;
    for (int i = threadIdx.x; i < N2; i += blockDim.x) Ds[i] = elem_D[e][i];
    __syncthreads();
    const double* Tm = trace + trace_offset[e];
    for (int f = 0; f < 6; ++f) {
      const SideDesc d = sd[6 * e + f];
      const int NQ = d.NQ, T = NQ * NQ;
      const double* Cm = face_ops + d.offC_m;
      const double* E = face_ops + d.offE;
      const double* g = geom + (size_t)7 * d.geom;
      // (-) side trace fields 0..3 -> A[0..3]; (+) side (re-ordered to the (-) ordering) -> A[4..7]
      for (int idx = threadIdx.x; idx < 4 * N2; idx += blockDim.x) A[(idx / N2) * fld_stride + idx % N2] = Tm[f * 4 * N2 + idx];
      if (d.kind != 0) {
        const int Np = d.Np, Np2 = Np * Np;
        const double* Tp = ((d.kind == 1) ? trace : ghost_trace) + d.nbr_trace + (size_t)d.f_p * 4 * Np2;
        for (int idx = threadIdx.x; idx < 4 * Np2; idx += blockDim.x) {
          const int c = idx / Np2, ab = idx % Np2;
          A[(4 + c) * fld_stride + ab] = Tp[c * Np2 + reorder_index(d.code, Np - 1, ab % Np, ab / Np)];
        }
      } else {
        // Dirichlet data g on the Lobatto face nodes plays the role of u_p (d4est_laplacian_flux_sipg.c:80-112)
        for (int idx = threadIdx.x; idx < N2; idx += blockDim.x) A[4 * fld_stride + idx] = bndry[d.bndry + idx];
      }
      __syncthreads();
      // to the mortar quadrature nodes: Bq[c] = (C (x) C) A[c]
      apply2d(Cm, NQ, N, A, tmp, Bq, 4, fld_stride, fld_stride, fld_stride);
      if (d.kind != 0) apply2d(face_ops + d.offC_p, NQ, d.Np, A + 4 * fld_stride, tmp, Bq + 4 * fld_stride, 4, fld_stride, fld_stride, fld_stride);
      else apply2d(Cm, NQ, N, A + 4 * fld_stride, tmp, Bq + 4 * fld_stride, 1, fld_stride, fld_stride, fld_stride);
      // SIPG terms at the quadrature nodes -> A[0]: term1 + term3 (both lifted without D^T), A[1..3]: term2_l
      for (int k = threadIdx.x; k < T; k += blockDim.x) {
        const double um = Bq[k], up = Bq[4 * fld_stride + k];
        double t1 = 0.0;
        double am[3];
        for (int i = 0; i < 3; ++i) {
          am[i] = g[i * T + k];
          t1 += am[i] * Bq[(1 + i) * fld_stride + k];
          if (d.kind != 0) t1 += g[(3 + i) * T + k] * Bq[(5 + i) * fld_stride + k];
        }
        const double jump = um - up;
        // interface: t1 = -1/2 sj n.(grad u_m + grad u_p), t2_l = -1/2 am_l [u]; boundary: t1 = -sj n.grad u_m, t2_l = -am_l (u - g)
        const double w1 = (d.kind != 0) ? -0.5 : -1.0;
        A[k] = w1 * t1 + g[6 * T + k] * jump;
        for (int l = 0; l < 3; ++l) A[(1 + l) * fld_stride + k] = w1 * am[l] * jump;
      }
      __syncthreads();
      // integrate + project onto the (-) side: Bq[c] (N x N) = (E (x) E) A[c]
      apply2d(E, N, NQ, A, tmp, Bq, 4, fld_stride, fld_stride, fld_stride);
      // lift (+ D_l^T for the term-2 fields) into the element accumulator
      const int dir = f >> 1, fix = face_fix(f, N);
      const int sdir = (dir == 0) ? 1 : (dir == 1 ? N : N2);
      for (int idx = threadIdx.x; idx < N3; idx += blockDim.x) {
        // node (pos along dir, a, b)
        const int pos = (idx / sdir) % N;
        int a, b;
        if (dir == 0) { a = (idx / N) % N; b = idx / N2; }
        else if (dir == 1) { a = idx % N; b = idx / N2; }
        else { a = idx % N; b = (idx / N) % N; }
        // normal direction: D^T lift of term2_dir touches every node of the line: D[fix][pos] * t2(a,b)
        double v = Ds[fix * N + pos] * Bq[(1 + dir) * fld_stride + a + N * b];
        if (pos == fix) {
          v += Bq[a + N * b];
          // tangential directions: (D^T)(a, a') within the face
          const int t0 = (dir == 0) ? 1 : 0, t1d = (dir == 2) ? 1 : 2;  // tangential reference directions, a <-> t0, b <-> t1d
          double s0 = 0.0, s1 = 0.0;
          for (int q = 0; q < N; ++q) {
            s0 = fma(Ds[q * N + a], Bq[(1 + t0) * fld_stride + q + N * b], s0);
            s1 = fma(Ds[q * N + b], Bq[(1 + t1d) * fld_stride + a + N * q], s1);
          }
          v += s0 + s1;
        }
        acc[idx] += v;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < N3; i += blockDim.x) Au[elem_ns[e] + i] += acc[i];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// (2b) fast flux kernel for plans whose sides all have N, Np, NQ <= 8 (p <= 7): one workgroup of six wavefronts per
// element, wavefront f <-> face f, lane <-> face node.  The 2-D tensor applies keep their results in registers, the
// four lifted fields (term1+3 and the three term2_l) are scattered into LDS volume fields W_0..W_3 in three
// conflict-free phases (opposite faces touch disjoint nodes), and one pass applies  W_0 + sum_l D_l^T W_l  and adds
// it to Au_e.  12 barriers per element instead of ~80 in the generic kernel.
// ---------------------------------------------------------------------------
constexpr int kFW = 8;  // max nodes per direction on the fast path

// All face arrays of the fast path live on a fixed 8 x 8 grid (index a + 8 b) and the operators are stored
// zero-padded to 8 x 8 in LDS, so the tensor applies are branch-free and fully unrolled for every p <= 7
// (the padding multiplies zeros).  out_c(a',b') = sum_{a,b} op[a'][a] op[b'][b] in_c(a,b); lane = a' + 8 b'.
template <int NF>
__device__ __forceinline__ void wave_apply2d(const double* op /*[8][8] padded*/, const double* in /*[NF][64]*/,
                                             double* tmp /*[NF][64]*/, int lane, double* out /*[NF]*/) {
  const int lo = lane & 7, hi = lane >> 3;
  double c[kFW];
  // pass 1: lane (a' = lo, b = hi): tmp_c(a', b) = sum_a op[a'][a] in_c(a, b)
#pragma unroll
  for (int a = 0; a < kFW; ++a) c[a] = op[lo * 8 + a];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < kFW; ++a) s = fma(c[a], in[f * 64 + a + 8 * hi], s);
    tmp[f * 64 + lane] = s;
  }
  __syncthreads();
  // pass 2: lane (a' = lo, b' = hi): out_c = sum_b op[b'][b] tmp_c(a', b)
#pragma unroll
  for (int b = 0; b < kFW; ++b) c[b] = op[hi * 8 + b];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < kFW; ++b) s = fma(c[b], tmp[f * 64 + lo + 8 * b], s);
    out[f] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(384) void flux_wave_kernel(const double* __restrict__ trace, const double* __restrict__ ghost_trace,
                                                        double* __restrict__ Au, const SideDesc* __restrict__ sd,
                                                        const ElemDesc* __restrict__ ed,
                                                        const double* __restrict__ face_ops, const double* __restrict__ geom,
                                                        const double* __restrict__ bndry, int n_elem) {
  __shared__ double s_in[6][4][64];   // per wave: 4 fields on the 8 x 8 grid
  __shared__ double s_tmp[6][4][64];
  __shared__ double s_ops[6][3][64];  // per wave: C_m, C_p, E of its side, zero-padded to 8 x 8
  __shared__ double s_W[4][512];      // lifted volume fields: W_0 (terms 1+3), W_1..3 (term 2_l)
  __shared__ double s_D[64];
  const int f = threadIdx.x >> 6;     // wave = face
  const int lane = threadIdx.x & 63;
  const int lo = lane & 7, hi = lane >> 3;
  double(*in)[64] = s_in[f];
  double(*tmp)[64] = s_tmp[f];
  // persistent workgroups: the descriptors of the NEXT element are requested while this one is computed
  int e = blockIdx.x;
  ElemDesc edn = ed[e < n_elem ? e : 0];
  SideDesc dn = sd[6 * (e < n_elem ? e : 0) + f];
  for (; e < n_elem; e += gridDim.x) {
    const ElemDesc el = edn;
    const SideDesc d = dn;
    {
      const int en = e + gridDim.x;
      if (en < n_elem) {
        edn = ed[en];
        dn = sd[6 * en + f];
      }
    }
    const int N = el.N, N2 = N * N, N3 = N2 * N;
    const int NQ = d.NQ, T = NQ * NQ;
    const int Np = d.Np, Np2 = Np * Np;
    const bool on_m = lo < N && hi < N, on_p = lo < Np && hi < Np, on_q = lo < NQ && hi < NQ;
    // ---- issue every global load of this side up front (independent requests: one memory latency)
    const double* Tm = trace + el.trace_off + (size_t)f * 4 * N2;
    double tm[4] = {0, 0, 0, 0}, tp[4] = {0, 0, 0, 0}, gq[7] = {0, 0, 0, 0, 0, 0, 0};
    if (on_m) {
#pragma unroll
      for (int c = 0; c < 4; ++c) tm[c] = Tm[c * N2 + lo + N * hi];
    }
    if (d.kind != 0) {
      const double* Tp = ((d.kind == 1) ? trace : ghost_trace) + d.nbr_trace + (size_t)d.f_p * 4 * Np2;
      if (on_p) {
        const int src = reorder_index(d.code, Np - 1, lo, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) tp[c] = Tp[c * Np2 + src];
      }
    } else if (on_m) {
      tp[0] = bndry[d.bndry + lo + N * hi];
    }
    if (on_q) {
      const double* g = geom + (size_t)7 * d.geom;
#pragma unroll
      for (int c = 0; c < 7; ++c) gq[c] = g[c * T + lo + NQ * hi];
    }
    // operators: entry (row hi, col lo) of the padded 8 x 8 images
    const double opm = (hi < NQ && lo < N) ? face_ops[d.offC_m + hi * N + lo] : 0.0;
    const double opp = (hi < NQ && lo < Np) ? face_ops[d.offC_p + hi * Np + lo] : 0.0;
    const double ope = (hi < N && lo < NQ) ? face_ops[d.offE + hi * NQ + lo] : 0.0;
    const double dval = (threadIdx.x < N2) ? face_ops[el.offD + threadIdx.x] : 0.0;
    // Au_e is read now (2 values per thread) so that the final update only stores
    double au0 = 0.0, au1 = 0.0;
    if ((int)threadIdx.x < N3) au0 = Au[el.ns + threadIdx.x];
    if ((int)threadIdx.x + 384 < N3) au1 = Au[el.ns + threadIdx.x + 384];
    for (int i = threadIdx.x; i < 4 * 512; i += blockDim.x) (&s_W[0][0])[i] = 0.0;
    s_ops[f][0][lane] = opm;
    s_ops[f][1][lane] = opp;
    s_ops[f][2][lane] = ope;
    if (threadIdx.x < N2) s_D[threadIdx.x] = dval;
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c][lane] = tm[c];
    __syncthreads();
    // ---- (-) side: 4 fields to the mortar quadrature nodes
    double qm[4], qp[4];
    wave_apply2d<4>(s_ops[f][0], &in[0][0], &tmp[0][0], lane, qm);
    // ---- (+) side (re-ordered to the (-) ordering) or the Dirichlet data (field 0 only; the others are zero)
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c][lane] = tp[c];
    __syncthreads();
    wave_apply2d<4>(s_ops[f][(d.kind != 0) ? 1 : 0], &in[0][0], &tmp[0][0], lane, qp);
    // ---- SIPG terms at the quadrature node of this lane (padding lanes carry zeros)
    {
      double t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) t1 += gq[i] * qm[1 + i] + gq[3 + i] * qp[1 + i];  // gq[3..5] = 0 on boundary sides
      const double jump = qm[0] - qp[0];
      const double w1 = (d.kind != 0) ? -0.5 : -1.0;
      in[0][lane] = w1 * t1 + gq[6] * jump;
#pragma unroll
      for (int l = 0; l < 3; ++l) in[1 + l][lane] = w1 * gq[l] * jump;
    }
    __syncthreads();
    // ---- integrate and project onto the (-) side: 4 fields on the N x N face nodes
    double res[4];
    wave_apply2d<4>(s_ops[f][2], &in[0][0], &tmp[0][0], lane, res);
    // ---- lift: scatter into the volume fields, opposite faces together (disjoint node sets)
    for (int phase = 0; phase < 3; ++phase) {
      if ((f >> 1) == phase && on_m) {
        const int v = face_vol_index(f, N, lo, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) s_W[c][v] += res[c];
      }
      __syncthreads();
    }
    // ---- Au_e += W_0 + sum_l D_l^T W_l
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int idx = threadIdx.x + rep * 384;
      if (idx < N3) {
        const int i = idx % N, j = (idx / N) % N, k = idx / N2;
        double v = s_W[0][idx];
        for (int q = 0; q < N; ++q) {
          v = fma(s_D[q * N + i], s_W[1][q + N * (j + N * k)], v);
          v = fma(s_D[q * N + j], s_W[2][i + N * (q + N * k)], v);
          v = fma(s_D[q * N + k], s_W[3][i + N * (j + N * q)], v);
        }
        Au[el.ns + idx] = (rep == 0 ? au0 : au1) + v;
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
  std::vector<int> elem_N, elem_ns, side_deg_m, side_deg_p;
  std::vector<const double*> elem_D;
  int* d_elem_N = nullptr;
  int* d_elem_ns = nullptr;
  int* d_side_deg_m = nullptr;
  int* d_side_deg_p = nullptr;
  const double** d_elem_D = nullptr;
  long long* d_ghost_trace_offset = nullptr;
  int* d_ghost_N = nullptr;
  int* d_ghost_ns = nullptr;
  const double** d_ghost_D = nullptr;
  std::map<int, double*> d_Dmat;  // per degree
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

void faces_setup(d4est_hip_plan* plan) {
  FaceHost& fh = g_face_host[plan];
  const int ne = plan->n_elements;
  const int qt = plan->quad_type;
  // per-degree derivative matrices
  auto get_D = [&](int deg) -> const double* {
    auto it = fh.d_Dmat.find(deg);
    if (it != fh.d_Dmat.end()) return it->second;
    double* d = upload_vec(Tables1D::dij(deg));
    fh.d_Dmat[deg] = d;
    return d;
  };
  fh.elem_N.resize(ne);
  fh.elem_ns.resize(ne);
  fh.elem_D.resize(ne);
  plan->trace_offset.resize(ne);
  long long toff = 0;
  int maxN = 1;
  for (int e = 0; e < ne; ++e) {
    const int N = plan->deg[e] + 1;
    fh.elem_N[e] = N;
    fh.elem_ns[e] = plan->nodal_stride[e];
    fh.elem_D[e] = get_D(plan->deg[e]);
    plan->trace_offset[e] = toff;
    toff += 24LL * N * N;
    maxN = std::max(maxN, N);
  }
  plan->local_trace_doubles = toff;
  plan->ghost_trace_offset.resize(plan->n_ghost);
  long long goff = 0;
  std::vector<int> ghost_N(plan->n_ghost), ghost_ns(plan->n_ghost);
  std::vector<const double*> ghost_D(plan->n_ghost);
  int gns = 0;
  for (int g = 0; g < plan->n_ghost; ++g) {
    const int N = plan->ghost_deg[g] + 1;
    plan->ghost_trace_offset[g] = goff;
    goff += 24LL * N * N;
    ghost_N[g] = N;
    ghost_ns[g] = gns;
    gns += N * N * N;
    ghost_D[g] = get_D(plan->ghost_deg[g]);
    maxN = std::max(maxN, N);
  }
  plan->ghost_trace_doubles = goff;

  // face operators, de-duplicated
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

  // derivative matrices inside the face-operator buffer (one load level less than a pointer table)
  std::map<int, int> offD_of;
  for (int e = 0; e < ne; ++e) {
    const int deg = plan->deg[e];
    if (!offD_of.count(deg)) {
      std::vector<double> D = Tables1D::dij(deg);
      offD_of[deg] = (int)ops.size();
      ops.insert(ops.end(), D.begin(), D.end());
    }
  }
  std::vector<ElemDesc> edv(ne);
  for (int e = 0; e < ne; ++e) {
    edv[e].N = plan->deg[e] + 1;
    edv[e].ns = plan->nodal_stride[e];
    edv[e].trace_off = plan->trace_offset[e];
    edv[e].offD = offD_of[plan->deg[e]];
    edv[e].pad = 0;
  }
  HIP_CHECK(hipMalloc(&plan->d_elem_desc, std::max<size_t>(edv.size(), 1) * sizeof(ElemDesc)));
  if (!edv.empty()) HIP_CHECK(hipMemcpy(plan->d_elem_desc, edv.data(), edv.size() * sizeof(ElemDesc), hipMemcpyHostToDevice));
  std::vector<SideDesc> sd(6 * (size_t)ne);
  fh.side_deg_m.assign(6 * (size_t)ne, 0);
  fh.side_deg_p.assign(6 * (size_t)ne, 0);
  int max_fld = maxN * maxN;
  for (int e = 0; e < ne; ++e)
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      SideDesc d{};
      const int nbr = plan->side_nbr[s];
      const int deg_m = plan->deg[e], degq_m = plan->deg_quad[e];
      int deg_p = deg_m, degq_p = degq_m;
      if (nbr == -1) {
        d.kind = 0;
      } else if (nbr >= 0) {
        if (nbr >= ne) D4EST_HIP_ABORT("plan_set_faces: side %zu neighbour %d out of range", s, nbr);
        d.kind = 1;
        deg_p = plan->deg[nbr];
        degq_p = plan->deg_quad[nbr];
        d.nbr_trace = plan->trace_offset[nbr];
      } else {
        const int g = -(nbr + 2);
        if (g >= plan->n_ghost) D4EST_HIP_ABORT("plan_set_faces: side %zu ghost %d out of range", s, g);
        d.kind = 2;
        deg_p = plan->ghost_deg[g];
        degq_p = plan->ghost_deg_quad[g];
        d.nbr_trace = plan->ghost_trace_offset[g];
      }
      const int deg_mq = std::max(degq_m, degq_p), deg_ml = std::max(deg_m, deg_p);
      d.f_p = plan->side_nbr_face[s];
      d.code = plan->side_reorder[s];
      d.Np = deg_p + 1;
      d.NQ = deg_mq + 1;
      d.offC_m = get_C(deg_m, deg_mq);
      d.offC_p = (d.kind == 0) ? d.offC_m : get_C(deg_p, deg_mq);
      d.offE = get_E(deg_m, d.kind == 0 ? deg_m : deg_ml, deg_mq);
      d.geom = plan->side_mortar_stride[s];
      d.bndry = plan->side_bndry_stride[s];
      sd[s] = d;
      fh.side_deg_m[s] = deg_m;
      fh.side_deg_p[s] = deg_p;
      max_fld = std::max(max_fld, std::max(d.NQ * d.NQ, std::max(d.NQ * d.Np, d.NQ * (deg_m + 1))));
    }
  plan->max_face_lds_doubles = 24 * max_fld + maxN * maxN * maxN + maxN * maxN;
  plan->face_fast = (max_fld <= 64);  // every N, Np, NQ <= 8
  if ((size_t)plan->max_face_lds_doubles * sizeof(double) > 160 * 1024) D4EST_HIP_ABORT("face kernel needs %d LDS doubles", plan->max_face_lds_doubles);

  static_assert(sizeof(SideDesc) % sizeof(int) == 0, "SideDesc layout");
  HIP_CHECK(hipMalloc(&plan->d_side_desc, std::max<size_t>(sd.size(), 1) * sizeof(SideDesc)));
  if (!sd.empty()) HIP_CHECK(hipMemcpy(plan->d_side_desc, sd.data(), sd.size() * sizeof(SideDesc), hipMemcpyHostToDevice));
  plan->d_trace_offset = upload_vec(plan->trace_offset);
  plan->d_face_ops = upload_vec(ops);
  fh.d_elem_N = upload_vec(fh.elem_N);
  fh.d_elem_ns = upload_vec(fh.elem_ns);
  fh.d_elem_D = upload_vec(fh.elem_D);
  fh.d_side_deg_m = upload_vec(fh.side_deg_m);
  fh.d_side_deg_p = upload_vec(fh.side_deg_p);
  fh.d_ghost_trace_offset = upload_vec(plan->ghost_trace_offset);
  fh.d_ghost_N = upload_vec(ghost_N);
  fh.d_ghost_ns = upload_vec(ghost_ns);
  fh.d_ghost_D = upload_vec(ghost_D);
  HIP_CHECK(hipMalloc(&plan->d_trace, std::max<size_t>((size_t)plan->local_trace_doubles, 1) * sizeof(double)));
  HIP_CHECK(hipMalloc(&plan->d_bndry, std::max<size_t>((size_t)plan->total_bndry_nodes, 1) * sizeof(double)));
  HIP_CHECK(hipMemset(plan->d_bndry, 0, std::max<size_t>((size_t)plan->total_bndry_nodes, 1) * sizeof(double)));
  HIP_CHECK(hipMalloc(&plan->d_face_geom, std::max<size_t>(7 * (size_t)plan->total_mortar_nodes, 1) * sizeof(double)));
  plan->has_faces = true;
  (void)max_fld;
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
  if (n_sides > 0) {
    const int grid = n_sides < 8192 ? n_sides : 8192;
    hipLaunchKernelGGL(face_geom_kernel, dim3(grid), dim3(64), 0, plan->stream, (const SideDesc*)plan->d_side_desc,
                       fh.d_side_deg_m, fh.d_side_deg_p, n_sides, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5],
                       plan->sipg_prefactor, plan->sipg_penalty_fcn, plan->d_face_geom);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  for (int i = 0; i < 6; ++i)
    if (tmp[i]) HIP_CHECK(hipFree(tmp[i]));
  plan->has_face_geometry = true;
}

void launch_traces(d4est_hip_plan* plan, const double* u, double* trace, bool ghost) {
  FaceHost& fh = g_face_host[plan];
  const int n = ghost ? plan->n_ghost : plan->n_elements;
  if (n == 0) return;
  int maxN = 1;
  if (ghost) for (int g = 0; g < n; ++g) maxN = std::max(maxN, plan->ghost_deg[g] + 1);
  else for (int e = 0; e < n; ++e) maxN = std::max(maxN, plan->deg[e] + 1);
  const size_t lds = ((size_t)maxN * maxN * maxN + (size_t)maxN * maxN) * sizeof(double);
  if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = n < 16384 ? n : 16384;
  hipLaunchKernelGGL(trace_kernel, dim3(grid), dim3(256), lds, plan->stream, u, trace, ghost ? fh.d_ghost_N : fh.d_elem_N,
                     ghost ? fh.d_ghost_ns : fh.d_elem_ns, ghost ? fh.d_ghost_trace_offset : plan->d_trace_offset,
                     ghost ? fh.d_ghost_D : fh.d_elem_D, n);
  HIP_CHECK(hipGetLastError());
}

void launch_flux(d4est_hip_plan* plan, const double* trace, const double* ghost_trace, double* Au) {
  FaceHost& fh = g_face_host[plan];
  if (!plan->has_faces || !plan->has_face_geometry) D4EST_HIP_ABORT("apply flux: plan_set_faces / plan_set_mortar_geometry were not called");
  if (plan->n_elements == 0) return;
  if (plan->n_ghost > 0 && !ghost_trace) D4EST_HIP_ABORT("apply flux: plan has %d ghost elements but no ghost trace buffer was given", plan->n_ghost);
  if (plan->face_fast && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0) {
    // persistent grid: 3 workgroups per CU are resident (LDS), each loops over elements
    const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
    const int resident = 3 * cus;
    const int rounds = (plan->n_elements + resident - 1) / resident;
    const int grid = (plan->n_elements + rounds - 1) / rounds;
    hipLaunchKernelGGL(flux_wave_kernel, dim3(grid), dim3(384), 0, plan->stream, trace, ghost_trace, Au,
                       (const SideDesc*)plan->d_side_desc, (const ElemDesc*)plan->d_elem_desc, plan->d_face_ops,
                       plan->d_face_geom, plan->d_bndry, plan->n_elements);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const size_t lds = (size_t)plan->max_face_lds_doubles * sizeof(double);
  if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(flux_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int fld_stride = (plan->max_face_lds_doubles > 0) ? 0 : 0;
  (void)fld_stride;
  // fld_stride recomputed exactly as in faces_setup
  int maxN = 1;
  for (int e = 0; e < plan->n_elements; ++e) maxN = std::max(maxN, plan->deg[e] + 1);
  for (int g = 0; g < plan->n_ghost; ++g) maxN = std::max(maxN, plan->ghost_deg[g] + 1);
  const int fs = (plan->max_face_lds_doubles - maxN * maxN * maxN - maxN * maxN) / 24;
  const int grid = plan->n_elements < 16384 ? plan->n_elements : 16384;
  hipLaunchKernelGGL(flux_kernel, dim3(grid), dim3(256), lds, plan->stream, trace, ghost_trace, Au,
                     (const SideDesc*)plan->d_side_desc, fh.d_elem_N, fh.d_elem_ns, plan->d_trace_offset, fh.d_elem_D,
                     plan->d_face_ops, plan->d_face_geom, plan->d_bndry, plan->n_elements, fs);
  HIP_CHECK(hipGetLastError());
}

void faces_destroy(d4est_hip_plan* plan) {
  auto it = g_face_host.find(plan);
  if (it != g_face_host.end()) {
    FaceHost& fh = it->second;
    (void)hipFree(fh.d_elem_N); (void)hipFree(fh.d_elem_ns); (void)hipFree(fh.d_elem_D);
    (void)hipFree(fh.d_side_deg_m); (void)hipFree(fh.d_side_deg_p);
    (void)hipFree(fh.d_ghost_trace_offset); (void)hipFree(fh.d_ghost_N); (void)hipFree(fh.d_ghost_ns); (void)hipFree(fh.d_ghost_D);
    for (auto& kv : fh.d_Dmat) (void)hipFree(kv.second);
    g_face_host.erase(it);
  }
  (void)hipFree(plan->d_elem_desc);
  (void)hipFree(plan->d_side_desc); (void)hipFree(plan->d_trace_offset); (void)hipFree(plan->d_face_ops);
  (void)hipFree(plan->d_face_geom); (void)hipFree(plan->d_bndry); (void)hipFree(plan->d_trace);
}

}  // namespace d4est_hip
