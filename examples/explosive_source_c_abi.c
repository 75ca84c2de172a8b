/* The explosive-source set-up of the reference (tests/explosive_source/explosive_source_lf4.py:7-56) driven through
 * the C-ABI alone - no Python, no torch: what a C / C++ / Fortran host (or the cgo / JNI / ctypes stub of
 * INTEGRATION.md) would write.  Plain C99; links against libseigen_hip.so only.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/explosive_source_c_abi.c -o build_tools/explosive_source_c_abi \
 *       -Lseigen_amd/csrc -lseigen_hip -Wl,-rpath,$PWD/seigen_amd/csrc -lm
 *   build_tools/explosive_source_c_abi [nx ny degree nsteps quadrilateral out.bin]
 *
 * Prints one line with the sizes, the device time per step and a checksum; writes the final velocity field
 * ([cell][node][2] doubles, the reference's layout) to out.bin if given.  tests/test_harness_gpu.py runs it and
 * compares the file bit for bit with the same run through the Python host layer. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "seigen_hip.h"

#define CHECK(call)                                                                        \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ != SG_OK) {                                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sg_last_error(h));               \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 120, ny = argc > 2 ? atoi(argv[2]) : 60;
  const int degree = argc > 3 ? atoi(argv[3]) : 2, nsteps = argc > 4 ? atoi(argv[4]) : 100;
  const int quad = argc > 5 ? atoi(argv[5]) : 0;
  const char* out_path = argc > 6 ? argv[6] : NULL;
  const double hcell = 2.5, Lx = nx * hcell, Ly = ny * hcell;   /* explosive_source_lf4.py:9-10 at h = 2.5 */
  sg_handle* h = NULL;

  sg_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.dim = 2;
  cfg.degree = degree;
  cfg.n[0] = nx;
  cfg.n[1] = ny;
  cfg.n[2] = 1;
  cfg.h[0] = cfg.h[1] = hcell;
  cfg.h[2] = 1.0;
  cfg.diagonal = quad ? 2 : 0;
  if (sg_create(&cfg, &h) != SG_OK) {
    fprintf(stderr, "sg_create: %s\n", sg_last_error(NULL));
    return 1;
  }
  sg_info_t info;
  CHECK(sg_get_info(h, &info));

  /* material and time step (:21-23, dt = 0.001 as uy.py:25: the CFL step of :30-32 is unstable with the sponge) */
  const double rho = 1.0, mu = 3600.0, lam = 3599.3664, dt = 1e-3;
  CHECK(sg_set_params(h, rho, dt, &lam, &mu, 0));

  /* DG4 sponge (:43-45): sigma = 1000 where x <= 20, x >= Lx - 20 or y <= 20, interpolated at the DG4 nodes */
  {
    const int nq = quad ? 25 : 15;
    const size_t nn = (size_t)info.ncells * nq;
    double* X = (double*)malloc(nn * 2 * sizeof(double));
    double* sigma = (double*)malloc(nn * sizeof(double));
    if (!X || !sigma) return 1;
    CHECK(sg_node_coords(h, 4, X, nn * 2 * sizeof(double)));
    for (size_t i = 0; i < nn; ++i)
      sigma[i] = (X[2 * i] <= 20.0 || X[2 * i] >= Lx - 20.0 || X[2 * i + 1] <= 20.0) ? 1000.0 : 0.0;
    CHECK(sg_set_absorption(h, sigma, 4));
    free(X);
    free(sigma);
  }

  /* source (:36-40): Ricker wavelet in the 1 m box one metre under the free surface, a = 159.42, delay 0.3 s;
   * here the delay is shortened to 0.03 s so that a short run sees the wavelet */
  {
    const double lo[2] = {44.5, Ly - 1.5}, hi[2] = {45.5, Ly - 0.5};
    CHECK(sg_set_source_box_ricker(h, lo, hi, 159.42, 0.03, dt, dt, nsteps));
  }

  /* zero initial conditions (:47-52) are the state of a fresh handle; run */
  CHECK(sg_step(h, nsteps));
  CHECK(sg_sync(h));
  double ms = 0.0;
  CHECK(sg_last_step_ms(h, &ms));

  const size_t nu = (size_t)info.u_dofs;
  double* u = (double*)malloc(nu * sizeof(double));
  if (!u) return 1;
  CHECK(sg_get_field(h, SG_FIELD_U, u, nu * sizeof(double)));
  double sum = 0.0, amax = 0.0;
  for (size_t i = 0; i < nu; ++i) {
    sum += u[i];
    if (fabs(u[i]) > amax) amax = fabs(u[i]);
  }
  printf("cells %lld nodes/cell %d dofs %lld steps %d device_ms_per_step %.5f max|u| %.6e sum(u) %.6e\n",
         (long long)info.ncells, (int)info.nd, (long long)(info.u_dofs + info.s_dofs), nsteps, ms / nsteps, amax, sum);
  if (out_path) {
    FILE* f = fopen(out_path, "wb");
    if (!f || fwrite(u, sizeof(double), nu, f) != nu) return 1;
    fclose(f);
  }
  free(u);
  sg_destroy(h);
  return 0;
}
