// Engine-owned 1-D operator tables (host side, C++).
//
// Replaces the lazily built per-degree cache of d4est_operators_t
// (reference: src/dGMath/d4est_operators.h:9-51, d4est_operators.c:196-304).
// The reference builds every table from Legendre Vandermonde matrices and
// LAPACK inversions; here the same operators are built from first principles
// with barycentric Lagrange formulas on the LGL nodes (no matrix inversion for
// D / interpolation / prolongation), which is the numerically stable form.
// All matrices are row-major rows x cols like the reference's.
#pragma once
#include <vector>

namespace d4est_hip {

enum QuadType { QUAD_LEGENDRE = 0, QUAD_LOBATTO = 1 };  // Quadrature/d4est_quadrature.h quadrature types "legendre"/"lobatto"

struct Tables1D {
  static constexpr int kMaxDeg = 23;  // reference tables stop at 20 LGL points (p <= 19); quadrature degree may exceed p

  // nodes / weights: n = deg + 1 points
  static void lobatto(int deg, std::vector<double>& x, std::vector<double>& w);  // GL_and_GLL_nodes_and_weights.h:4082
  static void gauss(int deg, std::vector<double>& x, std::vector<double>& w);    // GL_and_GLL_nodes_and_weights.h:6

  static std::vector<double> bary_weights(const std::vector<double>& x);
  // Lagrange interpolation matrix from nodes x (size n) to points y (size m): m x n
  static std::vector<double> interp_matrix(const std::vector<double>& x, const std::vector<double>& y);

  static std::vector<double> dij(int deg);                       // d4est_operators.c:855-872   (N x N)
  static std::vector<double> mij(int deg);                       // d4est_operators.c:712-724   (N x N)
  static std::vector<double> invmij(int deg);                    // d4est_operators.c:849-853
  static std::vector<double> lobatto_to_gauss(int deg, int deg_gauss);   // d4est_operators.c:411-438 (Ng x N)
  static std::vector<double> p_prolong(int degH, int degh);      // d4est_operators.c:995-1012  (Nh x NH)
  static std::vector<double> hp_prolong(int degH, int degh);     // d4est_operators.c:944-993   (2 x Nh x NH)
  static std::vector<double> p_restrict(int degH, int degh);     // d4est_operators.c:1165-1185 (NH x Nh)
  static std::vector<double> hp_restrict(int degH, int degh);    // d4est_operators.c:1232-1259 (2 x NH x Nh)

  // quadrature vtable equivalents (Quadrature/d4est_quadrature_legendre.c:22-93, _lobatto.c:23-93)
  static std::vector<double> quad_weights(int quad_type, int deg_quad);
  static std::vector<double> quad_interp(int quad_type, int deg, int deg_quad);  // Nq x N
  static std::vector<double> quad_diff(int quad_type, int deg_quad);   // Nq x Nq: differentiation matrix ON the quadrature nodes

  static std::vector<double> transpose(const std::vector<double>& A, int rows, int cols);
  static std::vector<double> matmul(const std::vector<double>& A, const std::vector<double>& B, int m, int l, int n);
  static bool invert(std::vector<double>& A, int n);
  // even-odd table of a centro-(anti)symmetric operator M (R x C, any parities): (C + 1) / 2 rows of R doubles, see
  // stiffness_wave_eo_kernel in d4est_hip_volume.hip and the comment in the definition
  static std::vector<double> eo_table(const std::vector<double>& M, int R, int C, bool antisymmetric);
};

}  // namespace d4est_hip
