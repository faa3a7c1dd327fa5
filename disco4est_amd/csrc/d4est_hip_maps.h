// Analytic tree maps evaluated ON THE DEVICE: x(tree, xi) for tree coordinates xi in [0,1]^3 and its
// Jacobian d x_i / d xi_j, for the geometries whose factors the engine can generate itself instead of receiving 96 B per
// quadrature node (volume) and 24 doubles per mortar node from the host.
//
//   D4EST_HIP_GEOM_CUBED_SPHERE_7TREE   [geometry] name = cubed_sphere_7tree: six wedges (trees 0..5) around a centre cube (tree 6),
//       d4est_geometry_cubed_sphere_7tree_X (src/Geometry/d4est_geometry_cubed_sphere.c:498-580), parameters R0, R1,
//       compactify_inner_shell; Clength = R0 / sqrt(3).  The reference evaluates DX from machine-generated closed forms
//       (:846-915, :1751-1830, GEOM_COMPUTE_ANALYTIC); here the Jacobian is the chain rule through the same map:
//       (a, b, c) = (2 xi0 - 1, 2 xi1 - 1, xi2 + 1), p = 2 - c, x = p a + (1-p) tan(pi a/4), y likewise,
//       S = 1 + (1-p)(tan^2 + tan^2) + 2 p, q = R(c) / sqrt(S), (X, Y, Z) = signed picks of (q x, q y, q) per wedge.
#pragma once
#include <hip/hip_runtime.h>

namespace d4est_hip {

struct TreeMapParams {
  int type;          // D4EST_HIP_GEOM_*
  int compactify;    // compactify_inner_shell
  double R0, R1, Clength;
};

__host__ __device__ inline void cubed_sphere7_dxdxi(const TreeMapParams& P, int tree, const double xi[3], double D[3][3]) {
  if (tree == 6) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) D[i][j] = (i == j) ? 2.0 * P.Clength : 0.0;
    return;
  }
  const double kPi4 = 0.78539816339744830962;
  const double a = 2.0 * xi[0] - 1.0, b = 2.0 * xi[1] - 1.0, c = xi[2] + 1.0;
  double R, dR;
  if (P.compactify) {
    const double m = 1.0 / (1.0 / P.R1 - 1.0 / P.R0), t = (P.R0 - 2.0 * P.R1) / (P.R0 - P.R1);
    R = m / (c - t);
    dR = -m / ((c - t) * (c - t));
  } else {
    R = P.R0 * (2.0 - c) + P.R1 * (c - 1.0);
    dR = P.R1 - P.R0;
  }
  const double p = 2.0 - c;
  const double tx = tan(a * kPi4), ty = tan(b * kPi4);
  const double dtx = kPi4 * (1.0 + tx * tx), dty = kPi4 * (1.0 + ty * ty);
  const double x = p * a + (1.0 - p) * tx, y = p * b + (1.0 - p) * ty;
  const double S = 1.0 + (1.0 - p) * (tx * tx + ty * ty) + 2.0 * p;
  const double rs = 1.0 / sqrt(S), q = R * rs;
  // derivatives with respect to (a, b, c); dp/dc = -1
  const double dx[3] = {p + (1.0 - p) * dtx, 0.0, tx - a};
  const double dy[3] = {0.0, p + (1.0 - p) * dty, ty - b};
  const double dS[3] = {(1.0 - p) * 2.0 * tx * dtx, (1.0 - p) * 2.0 * ty * dty, (tx * tx + ty * ty) - 2.0};
  const double h = -0.5 * R * rs * rs * rs;
  const double dq[3] = {h * dS[0], h * dS[1], dR * rs + h * dS[2]};
  const double sc[3] = {2.0, 2.0, 1.0};   // d(a, b, c) / d xi
  double g[3][3];                         // rows: q x, q y, q
  for (int k = 0; k < 3; ++k) {
    g[0][k] = (dq[k] * x + q * dx[k]) * sc[k];
    g[1][k] = (dq[k] * y + q * dy[k]) * sc[k];
    g[2][k] = dq[k] * sc[k];
  }
  // wedge -> (X, Y, Z): d4est_geometry_cubed_sphere.c:543-577
  const int pick[6][3] = {{0, 2, 1}, {0, 1, 2}, {0, 2, 1}, {2, 0, 1}, {1, 0, 2}, {2, 0, 1}};
  const double sign[6][3] = {{1, -1, 1}, {1, 1, 1}, {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, -1, 1}};
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) D[i][k] = sign[tree][i] * g[pick[tree][i]][k];
}

__host__ __device__ inline void tree_map_dxdxi(const TreeMapParams& P, int tree, const double xi[3], double D[3][3]) {
  cubed_sphere7_dxdxi(P, tree, xi, D);
}

// inverse and determinant of a 3 x 3 matrix
__host__ __device__ inline double invert3(const double A[3][3], double I[3][3]) {
  const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
  const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
  const double r = 1.0 / det;
  I[0][0] = c00 * r; I[1][0] = c01 * r; I[2][0] = c02 * r;
  I[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r;
  I[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r;
  I[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r;
  I[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
  I[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
  I[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
  return det;
}

// a (possibly virtual) cell of the forest: tree, corner q (p4est integer coordinates), side dq
struct CellDesc {
  int tree, q[3], dq, face;
};

// d x / d r (element reference coordinates r in [-1,1]^3) of `cell` at reference point r
__host__ __device__ inline void cell_dxdr(const TreeMapParams& P, const CellDesc& cell, double root_len, const double r[3], double dxdr[3][3]) {
  double xi[3];
  for (int d = 0; d < 3; ++d) xi[d] = ((double)cell.q[d] + 0.5 * (double)cell.dq * (r[d] + 1.0)) / root_len;
  double D[3][3];
  tree_map_dxdxi(P, cell.tree, xi, D);
  const double s = 0.5 * (double)cell.dq / root_len;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dxdr[i][j] = D[i][j] * s;
}

}  // namespace d4est_hip
