/* Plain C99 host of the C-ABI (include/d4est_hip.h): no Python, no C++, no torch.
 *
 * Reproduces the outputs of the REFERENCE's d4est_quadrature_apply_stiffness_matrix recorded at survey time (SURVEY.md Appendix A,
 * tests/golden/survey_probe.json): one affine element, h = 1/8, u = x^2 + 2 y^2 + 3 z^2 + x y z at the Lobatto nodes, deg_quad = deg,
 * Gauss-Legendre, rst_xyz[i][j] = delta_ij 2/h, J = h^3/8 -- the way a d4est build would drive the library: host arrays in the
 * reference's layout, device buffers from d4est_hip_malloc, one plan, one apply.
 *
 * Build / run (tests/test_c_abi_gpu.py does this):  gcc -std=c99 -O2 -Iinclude tests/c/abi_probe.c -Ldisco4est_amd -ld4est_hip -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "d4est_hip.h"

static int run_case(int p, double want_sq, double want_0) {
  const double h = 0.125;
  const int N = p + 1, n3 = N * N * N;
  int deg = p, deg_quad = p, nodal_stride = 0, quad_stride = 0;
  double* x1 = (double*)malloc(sizeof(double) * N);
  if (d4est_hip_table(D4EST_HIP_TABLE_LOBATTO_NODES, p, 0, x1) != N) return 1;
  double* u = (double*)malloc(sizeof(double) * n3);
  double* Au = (double*)malloc(sizeof(double) * n3);
  for (int k = 0; k < N; k++)
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        const double x = h * (x1[i] + 1) / 2, y = h * (x1[j] + 1) / 2, z = h * (x1[k] + 1) / 2;
        u[i + N * (j + N * k)] = x * x + 2 * y * y + 3 * z * z + x * y * z;   /* x fastest, d4est_operators.c:1318-1323 */
      }
  /* geometric factors in the reference's SoA layout (Mesh/d4est_mesh.c:2757-2776) */
  double* J = (double*)malloc(sizeof(double) * n3);
  double* rst = (double*)calloc((size_t)9 * n3, sizeof(double));
  for (int n = 0; n < n3; n++) {
    J[n] = h * h * h / 8;
    for (int i = 0; i < 3; i++) rst[(size_t)(3 * i + i) * n3 + n] = 2 / h;
  }
  d4est_hip_plan_t* plan = d4est_hip_plan_create(1, &deg, &deg_quad, &nodal_stride, &quad_stride, D4EST_HIP_QUAD_LEGENDRE);
  if (d4est_hip_plan_local_nodes(plan) != n3) return 2;
  d4est_hip_plan_set_geometry(plan, J, rst, 0 /* host arrays */);
  double* u_dev = (double*)d4est_hip_malloc(sizeof(double) * n3);
  double* Au_dev = (double*)d4est_hip_malloc(sizeof(double) * n3);
  d4est_hip_memcpy_h2d(u_dev, u, sizeof(double) * n3);
  d4est_hip_apply_stiffness_matrix(plan, u_dev, Au_dev);
  d4est_hip_device_synchronize();
  d4est_hip_memcpy_d2h(Au, Au_dev, sizeof(double) * n3);
  double sq = 0, sum = 0;
  for (int n = 0; n < n3; n++) { sq += Au[n] * Au[n]; sum += Au[n]; }
  const double e_sq = fabs(sq - want_sq) / want_sq, e_0 = fabs(Au[0] - want_0) / fabs(want_0);
  printf("p=%2d  |Au|^2 = %.15e (reference %.15e, rel %.1e)  Au[0] = %.15e (reference %.15e, rel %.1e)  sum Au = %.1e  kernel %s\n", p, sq,
         want_sq, e_sq, Au[0], want_0, e_0, sum, d4est_hip_plan_last_kernel(plan));
  d4est_hip_free(u_dev); d4est_hip_free(Au_dev);
  d4est_hip_plan_destroy(plan);
  free(x1); free(u); free(Au); free(J); free(rst);
  return (e_sq <= 1e-11 && e_0 <= 1e-9 && fabs(sum) <= 1e-12) ? 0 : 3;   /* Au[0] is 8 orders below u: looser bound */
}

int main(void) {
  if (d4est_hip_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 77; }
  printf("%s\n", d4est_hip_version());
  int rc = 0;
  rc |= run_case(3, 4.300465922296779e-05, -1.356336805555552e-05);
  rc |= run_case(7, 7.965107281075192e-06, -1.334587964650292e-07);
  rc |= run_case(11, 3.108086730277706e-06, -1.019035917020307e-08);
  rc |= run_case(15, 1.629628276226040e-06, -1.695421006951411e-09);
  printf(rc ? "MISMATCH\n" : "ok\n");
  return rc;
}
