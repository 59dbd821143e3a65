/* adv_1d.c - a C driver on the airhip C ABI, shaped like the reference's tests/adv_1d.c:
 * 1D upwind advection (A[0,0]=1; A[i,i-1]=-1, A[i,i]=1), b = 0, x0 = 1, right-preconditioned
 * GMRES(30) with PCAIR (power basis, a_drop 1e-3, matrix-free Newton coarse solver: the options of
 * tests/Makefile:537-540).  Exit status 0 iff the solve converged within -ksp_max_it (default 2),
 * exactly the reference's pass criterion (tests/adv_1d.c:195-199).
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I../include -I/opt/rocm/include adv_1d.c -o adv_1d \
 *       -L../pflare_amd -lairhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/../pflare_amd
 *   ./adv_1d [-n 1000] [-ksp_max_it 2]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "airhip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, airhip_last_error()); return 1; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  long n = 1000; int max_it = 2;
  for (int i = 1; i + 1 < argc; i++) {
    if (!strcmp(argv[i], "-n")) n = atol(argv[i + 1]);
    if (!strcmp(argv[i], "-ksp_max_it")) max_it = atoi(argv[i + 1]);
  }
  airhip_hostmat A; airhip_csr csr; airhip_pcair pc;
  CHECK(airhip_gen_adv_1d(n, &A));
  CHECK(airhip_hostmat_get(A, &csr));
  CHECK(airhip_pcair_create(&pc));
  CHECK(airhip_pcair_set_option(pc, "-pc_air_a_drop", "1e-3"));
  CHECK(airhip_pcair_set_option(pc, "-pc_air_inverse_type", "power"));
  CHECK(airhip_pcair_set_option(pc, "-pc_air_coarsest_inverse_type", "newton"));
  CHECK(airhip_pcair_set_option(pc, "-pc_air_coarsest_matrix_free_polys", "1"));
  CHECK(airhip_pcair_set_option(pc, "-pc_air_coarsest_poly_order", "10"));
  CHECK(airhip_pcair_setup(pc, &csr));                  /* PCSetUp */

  double *b, *x, *hx = (double*)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) hx[i] = 1.0;             /* x0 = 1 (tests/adv_1d.c:160) */
  HIPCHECK(hipMalloc((void**)&b, sizeof(double) * n));
  HIPCHECK(hipMalloc((void**)&x, sizeof(double) * n));
  HIPCHECK(hipMemset(b, 0, sizeof(double) * n));        /* b = 0 (tests/adv_1d.c:118) */
  HIPCHECK(hipMemcpy(x, hx, sizeof(double) * n, hipMemcpyHostToDevice));

  airhip_ksp_options o; airhip_ksp_result res; int nh = 0;
  airhip_ksp_default_options(&o);
  o.type = 1; o.pc_side = 1; o.rtol = 1e-10; o.atol = 1e-50; o.max_it = max_it;   /* gmres, right PC */
  CHECK(airhip_ksp_solve(&o, NULL, pc, NULL, b, x, &res, NULL, 0, &nh));          /* KSPSolve */
  HIPCHECK(hipMemcpy(hx, x, sizeof(double) * n, hipMemcpyDeviceToHost));
  double mx = 0.0;
  for (long i = 0; i < n; i++) { double a = hx[i] < 0 ? -hx[i] : hx[i]; if (a > mx) mx = a; }
  int levels = 0;
  CHECK(airhip_pcair_num_levels(pc, &levels));
  printf("adv_1d n=%ld levels=%d iterations=%d reason=%d rnorm=%.3e max|x|=%.3e\n", n, levels, res.its, res.reason, res.rnorm, mx);

  hipFree(b); hipFree(x); free(hx);
  CHECK(airhip_pcair_destroy(&pc));
  CHECK(airhip_hostmat_destroy(&A));
  return res.reason > 0 ? 0 : 1;
}
