// Engine-owned 1-D operator tables; see d4est_hip_tables.h.
#include "d4est_hip_tables.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace d4est_hip {

namespace {

// Legendre P_n and P_n' (three-term recurrence), long double for the Newton solves.
void legendre(int n, long double x, long double& p, long double& dp) {
  long double p0 = 1.0L, p1 = x;
  if (n == 0) { p = 1.0L; dp = 0.0L; return; }
  for (int k = 2; k <= n; ++k) {
    long double pk = ((2.0L * k - 1.0L) * x * p1 - (k - 1.0L) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  p = p1;
  if (fabsl(x) == 1.0L) dp = 0.5L * n * (n + 1.0L) * ((x > 0 || (n % 2 == 1)) ? 1.0L : -1.0L);
  else dp = n * (x * p1 - p0) / (x * x - 1.0L);
}

void check_deg(int deg) {
  if (deg < 1 || deg > Tables1D::kMaxDeg) {
    std::fprintf(stderr, "[D4EST_HIP_ABORT] degree %d outside [1,%d]\n", deg, Tables1D::kMaxDeg);
    std::abort();
  }
}

}  // namespace

void Tables1D::gauss(int deg, std::vector<double>& x, std::vector<double>& w) {
  check_deg(deg);
  const int n = deg + 1;
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const long double pi = acosl(-1.0L);
  for (int i = 0; i < (n + 1) / 2; ++i) {
    long double xi = -cosl(pi * (i + 0.75L) / (n + 0.5L));
    long double p, dp;
    for (int it = 0; it < 100; ++it) {
      legendre(n, xi, p, dp);
      long double dx = p / dp;
      xi -= dx;
      if (fabsl(dx) < 1e-19L) break;
    }
    legendre(n, xi, p, dp);
    long double wi = 2.0L / ((1.0L - xi * xi) * dp * dp);
    x[i] = (double)xi;
    x[n - 1 - i] = (double)(-xi);
    w[i] = w[n - 1 - i] = (double)wi;
  }
  if (n % 2) x[n / 2] = 0.0;
}

void Tables1D::lobatto(int deg, std::vector<double>& x, std::vector<double>& w) {
  check_deg(deg);
  const int n = deg + 1, N = deg;
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const long double pi = acosl(-1.0L);
  std::vector<long double> xl(n);
  xl[0] = -1.0L;
  xl[N] = 1.0L;
  for (int i = 1; i <= N / 2; ++i) {
    long double xi = -cosl(pi * i / N);
    for (int it = 0; it < 100; ++it) {
      long double p, dp;
      legendre(N, xi, p, dp);
      long double ddp = (2.0L * xi * dp - N * (N + 1.0L) * p) / (1.0L - xi * xi);
      long double dx = dp / ddp;
      xi -= dx;
      if (fabsl(dx) < 1e-19L) break;
    }
    xl[i] = xi;
    xl[N - i] = -xi;
  }
  if (n % 2) xl[n / 2] = 0.0L;
  for (int i = 0; i < n; ++i) {
    long double p, dp;
    legendre(N, xl[i], p, dp);
    x[i] = (double)xl[i];
    w[i] = (double)(2.0L / (N * (N + 1.0L) * p * p));
  }
  // Opt-in bit-compatibility with the reference's TABULATED nodes: its Lobatto table for n = 12 (p = 11, BASELINE config 3) carries a
  // digit slip in one abscissa pair, +-0.6328761530318697 for the root 0.63287615303186067766... of P_11'
  // (src/dGMath/GL_and_GLL_nodes_and_weights.h:4327,4332; 9e-15 off).  By default the engine uses the root; with
  // D4EST_HIP_REFERENCE_NODE_TABLES=1 it uses the reference's number, and every table derived from the nodes (D, V, the interpolation
  // and transfer operators) follows.  tests/test_bench_size_gpu.py bounds what the difference does to A u.
  static const bool ref_tables = []{ const char* e = std::getenv("D4EST_HIP_REFERENCE_NODE_TABLES"); return e && e[0] == '1'; }();
  if (ref_tables && n == 12) {
    x[3] = -0.6328761530318697;
    x[8] = 0.6328761530318697;
  }
}

std::vector<double> Tables1D::bary_weights(const std::vector<double>& x) {
  const int n = (int)x.size();
  std::vector<double> lam(n);
  for (int j = 0; j < n; ++j) {
    long double prod = 1.0L;
    for (int k = 0; k < n; ++k)
      if (k != j) prod *= ((long double)x[j] - (long double)x[k]);
    lam[j] = (double)(1.0L / prod);
  }
  return lam;
}

std::vector<double> Tables1D::interp_matrix(const std::vector<double>& x, const std::vector<double>& y) {
  const int n = (int)x.size(), m = (int)y.size();
  std::vector<double> lam = bary_weights(x);
  std::vector<double> I((size_t)m * n, 0.0);
  for (int a = 0; a < m; ++a) {
    int hit = -1;
    for (int j = 0; j < n; ++j)
      if (y[a] == x[j]) hit = j;
    if (hit >= 0) {
      I[(size_t)a * n + hit] = 1.0;
      continue;
    }
    long double den = 0.0L;
    for (int j = 0; j < n; ++j) den += (long double)lam[j] / ((long double)y[a] - (long double)x[j]);
    for (int j = 0; j < n; ++j)
      I[(size_t)a * n + j] = (double)(((long double)lam[j] / ((long double)y[a] - (long double)x[j])) / den);
  }
  return I;
}

std::vector<double> Tables1D::dij(int deg) {
  std::vector<double> x, w;
  lobatto(deg, x, w);
  const int n = deg + 1;
  std::vector<double> lam = bary_weights(x);
  std::vector<double> D((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    long double diag = 0.0L;
    for (int j = 0; j < n; ++j) {
      if (i == j) continue;
      long double d = ((long double)lam[j] / (long double)lam[i]) / ((long double)x[i] - (long double)x[j]);
      D[(size_t)i * n + j] = (double)d;
      diag -= d;
    }
    D[(size_t)i * n + i] = (double)diag;
  }
  return D;
}

// Differentiation matrix of the Lagrange basis on the quadrature nodes (barycentric form, as dij on the Lobatto nodes).  With deg_quad =
// deg it equals B D B^-1 (B = quad_interp, square): the derivative of the interpolant evaluated where it was interpolated to -- the
// collocated-gradient form of the stiffness apply (stiffness_mw_element_cg).
std::vector<double> Tables1D::quad_diff(int quad_type, int deg_quad) {
  std::vector<double> x, w;
  if (quad_type == QUAD_LEGENDRE) gauss(deg_quad, x, w);
  else if (quad_type == QUAD_LOBATTO) lobatto(deg_quad, x, w);
  else { std::fprintf(stderr, "[D4EST_HIP_ABORT] unknown quadrature type %d\n", quad_type); std::abort(); }
  const int n = deg_quad + 1;
  std::vector<double> lam = bary_weights(x);
  std::vector<double> D((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    long double diag = 0.0L;
    for (int j = 0; j < n; ++j) {
      if (i == j) continue;
      long double d = ((long double)lam[j] / (long double)lam[i]) / ((long double)x[i] - (long double)x[j]);
      D[(size_t)i * n + j] = (double)d;
      diag -= d;
    }
    D[(size_t)i * n + i] = (double)diag;
  }
  return D;
}

std::vector<double> Tables1D::transpose(const std::vector<double>& A, int rows, int cols) {
  std::vector<double> At((size_t)rows * cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) At[(size_t)j * rows + i] = A[(size_t)i * cols + j];
  return At;
}

std::vector<double> Tables1D::matmul(const std::vector<double>& A, const std::vector<double>& B, int m, int l, int n) {
  std::vector<double> C((size_t)m * n, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      long double s = 0.0L;
      for (int k = 0; k < l; ++k) s += (long double)A[(size_t)i * l + k] * (long double)B[(size_t)k * n + j];
      C[(size_t)i * n + j] = (double)s;
    }
  return C;
}

bool Tables1D::invert(std::vector<double>& A, int n) {
  std::vector<long double> M((size_t)n * 2 * n, 0.0L);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) M[(size_t)i * 2 * n + j] = A[(size_t)i * n + j];
    M[(size_t)i * 2 * n + n + i] = 1.0L;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    long double best = fabsl(M[(size_t)c * 2 * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (fabsl(M[(size_t)r * 2 * n + c]) > best) { best = fabsl(M[(size_t)r * 2 * n + c]); piv = r; }
    if (best == 0.0L) return false;
    if (piv != c)
      for (int j = 0; j < 2 * n; ++j) std::swap(M[(size_t)c * 2 * n + j], M[(size_t)piv * 2 * n + j]);
    long double d = 1.0L / M[(size_t)c * 2 * n + c];
    for (int j = 0; j < 2 * n; ++j) M[(size_t)c * 2 * n + j] *= d;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      long double f = M[(size_t)r * 2 * n + c];
      if (f == 0.0L) continue;
      for (int j = 0; j < 2 * n; ++j) M[(size_t)r * 2 * n + j] -= f * M[(size_t)c * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (double)M[(size_t)i * 2 * n + n + j];
  return true;
}

// Exact 1-D mass matrix M_ij = int l_i l_j: the integrand has degree 2p, so the
// (p+1)-point Gauss rule (exact to degree 2p+1) integrates it exactly.
std::vector<double> Tables1D::mij(int deg) {
  const int n = deg + 1;
  std::vector<double> xg, wg;
  gauss(deg, xg, wg);
  std::vector<double> Bg = lobatto_to_gauss(deg, deg);
  std::vector<double> M((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      long double s = 0.0L;
      for (int q = 0; q < n; ++q) s += (long double)wg[q] * (long double)Bg[(size_t)q * n + i] * (long double)Bg[(size_t)q * n + j];
      M[(size_t)i * n + j] = (double)s;
    }
  return M;
}

std::vector<double> Tables1D::invmij(int deg) {
  std::vector<double> M = mij(deg);
  if (!invert(M, deg + 1)) { std::fprintf(stderr, "[D4EST_HIP_ABORT] singular mass matrix\n"); std::abort(); }
  return M;
}

std::vector<double> Tables1D::lobatto_to_gauss(int deg, int deg_gauss) {
  std::vector<double> x, w, xg, wg;
  lobatto(deg, x, w);
  gauss(deg_gauss, xg, wg);
  return interp_matrix(x, xg);
}

std::vector<double> Tables1D::p_prolong(int degH, int degh) {
  std::vector<double> xH, wH, xh, wh;
  lobatto(degH, xH, wH);
  lobatto(degh, xh, wh);
  return interp_matrix(xH, xh);
}

std::vector<double> Tables1D::hp_prolong(int degH, int degh) {
  std::vector<double> xH, wH, xh, wh;
  lobatto(degH, xH, wH);
  lobatto(degh, xh, wh);
  const int nh = degh + 1, nH = degH + 1;
  std::vector<double> P2((size_t)2 * nh * nH);
  for (int c = 0; c < 2; ++c) {
    std::vector<double> y(nh);
    for (int i = 0; i < nh; ++i) y[i] = 0.5 * xh[i] + (c == 0 ? -0.5 : 0.5);  // d4est_reference.c:36-47
    std::vector<double> P = interp_matrix(xH, y);
    for (size_t i = 0; i < P.size(); ++i) P2[(size_t)c * nh * nH + i] = P[i];
  }
  return P2;
}

// R = M_H^{-1} P^T M_h  (d4est_operators.c:1134-1185: the 0.5 inside the aux routine is undone by the final x2)
std::vector<double> Tables1D::p_restrict(int degH, int degh) {
  const int nh = degh + 1, nH = degH + 1;
  std::vector<double> P = p_prolong(degH, degh);
  std::vector<double> Pt = transpose(P, nh, nH);
  std::vector<double> PtMh = matmul(Pt, mij(degh), nH, nh, nh);
  return matmul(invmij(degH), PtMh, nH, nH, nh);
}

// per child: R_c = M_H^{-1} (0.5 P_c)^T M_h  (d4est_operators.c:1232-1259)
std::vector<double> Tables1D::hp_restrict(int degH, int degh) {
  const int nh = degh + 1, nH = degH + 1;
  std::vector<double> P2 = hp_prolong(degH, degh);
  std::vector<double> Mh = mij(degh), iMH = invmij(degH);
  std::vector<double> R2((size_t)2 * nh * nH);
  for (int c = 0; c < 2; ++c) {
    std::vector<double> P(P2.begin() + (size_t)c * nh * nH, P2.begin() + (size_t)(c + 1) * nh * nH);
    std::vector<double> Pt = transpose(P, nh, nH);
    for (auto& v : Pt) v *= 0.5;
    std::vector<double> R = matmul(iMH, matmul(Pt, Mh, nH, nh, nh), nH, nH, nh);
    for (size_t i = 0; i < R.size(); ++i) R2[(size_t)c * nh * nH + i] = R[i];
  }
  return R2;
}

std::vector<double> Tables1D::quad_weights(int quad_type, int deg_quad) {
  std::vector<double> x, w;
  if (quad_type == QUAD_LEGENDRE) gauss(deg_quad, x, w);
  else if (quad_type == QUAD_LOBATTO) lobatto(deg_quad, x, w);
  else { std::fprintf(stderr, "[D4EST_HIP_ABORT] unknown quadrature type %d\n", quad_type); std::abort(); }
  return w;
}

std::vector<double> Tables1D::eo_table(const std::vector<double>& M, int R, int C, bool antisymmetric) {
  // (C + 1) / 2 rows of R doubles.  Row c < C / 2 serves the column pair (c, C - 1 - c): entries [0, (R + 1) / 2) multiply the FIRST
  // combination of the pair (x_c + x_{C-1-c} for a symmetric M, x_c - x_{C-1-c} for an antisymmetric one) and give a_r (plus, for odd R,
  // the middle output itself at index R / 2); entries [(R + 1) / 2, R) multiply the other combination and give b_r;
  // y_r = a_r + b_r, y_{R-1-r} = a_r - b_r.  For odd C the last row serves the middle column (its input enters both parts as it is).
  const int hc = C / 2, hr = R / 2, rows = (C + 1) / 2, split = (R + 1) / 2;
  std::vector<double> T((size_t)rows * R, 0.0);
  for (int c = 0; c < hc; ++c) {
    for (int r = 0; r < hr; ++r) {
      const double me = 0.5 * (M[(size_t)r * C + c] + M[(size_t)r * C + (C - 1 - c)]);
      const double mo = 0.5 * (M[(size_t)r * C + c] - M[(size_t)r * C + (C - 1 - c)]);
      T[(size_t)c * R + r] = antisymmetric ? mo : me;
      T[(size_t)c * R + split + r] = antisymmetric ? me : mo;
    }
    if (R % 2) T[(size_t)c * R + hr] = M[(size_t)hr * C + c];   // middle output row: sum_c M[hr][c] (x_c +- x_{C-1-c})
  }
  if (C % 2) {
    for (int r = 0; r < hr; ++r) {
      if (antisymmetric) T[(size_t)hc * R + split + r] = M[(size_t)r * C + hc];
      else T[(size_t)hc * R + r] = M[(size_t)r * C + hc];
    }
    if (R % 2 && !antisymmetric) T[(size_t)hc * R + hr] = M[(size_t)hr * C + hc];   // (zero for an antisymmetric M)
  }
  return T;
}

std::vector<double> Tables1D::quad_interp(int quad_type, int deg, int deg_quad) {
  if (quad_type == QUAD_LEGENDRE) return lobatto_to_gauss(deg, deg_quad);
  if (quad_type == QUAD_LOBATTO) return p_prolong(deg, deg_quad);
  std::fprintf(stderr, "[D4EST_HIP_ABORT] unknown quadrature type %d\n", quad_type);
  std::abort();
}

}  // namespace d4est_hip
