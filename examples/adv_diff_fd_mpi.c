/* adv_diff_fd_mpi.c - an MPI + C driver on the airhip C ABI, shaped like the reference's tests/adv_diff_fd.c run as
 * `mpiexec -n N ./adv_diff_fd -pc_type air ...` (tests/Makefile:1121-1133 and the -n 2 commands): the finite-difference
 * advection-diffusion operator of tests/adv_diff_fd.c:366-586 distributed in contiguous row blocks (MPIAIJ ownership),
 * b = 0, x0 = 1, right-preconditioned GMRES(30) with PCAIR.  Every rank drives its own GPU (or shares device 0 with
 * -same_device 1, as the one-GPU test box requires; the halo then travels over MPI through host buffers).
 * Exit status 0 iff the solve converged within -ksp_max_it: the reference's pass criterion (tests/adv_diff_fd.c:342-347).
 *
 *   make -C pflare_amd/csrc mpi && make -C examples adv_diff_fd_mpi
 *   mpiexec -n 2 examples/adv_diff_fd_mpi -dim 3 -da_grid_x 24 -pc_air_a_lump 1 -ksp_max_it 6
 * Options: -dim 2|3, -da_grid_x/-da_grid_y/-da_grid_z, -u -v -w -alpha, -ksp_max_it, -ksp_rtol, -transport staged|rccl,
 * -same_device 0|1, -layout_hint 0|1 (brick keys for the device layout, default 1), -host_only 1 (stop after every rank has
 * received its piece: no GPU needed), any -pc_air_<name> <value>.
 */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "airhip.h"
#include "airhip_mpi.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); MPI_Abort(MPI_COMM_WORLD, 1); } while (0)

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  int dim = 3, max_it = 10, host_only = 0, same_device = 0, transport = AIRHIP_MPI_TRANSPORT_STAGED, use_hint = 1;
  long gx = 16, gy = -1, gz = -1;
  double u = 1.0, v = 1.0, w = 1.0, alpha = 0.0, rtol = 1e-10;
  const char* names[64]; const char* values[64]; int nopts = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    const char* k = argv[i]; const char* val = argv[i + 1];
    if (!strcmp(k, "-dim")) dim = atoi(val);
    else if (!strcmp(k, "-da_grid_x")) gx = atol(val);
    else if (!strcmp(k, "-da_grid_y")) gy = atol(val);
    else if (!strcmp(k, "-da_grid_z")) gz = atol(val);
    else if (!strcmp(k, "-u")) u = atof(val);
    else if (!strcmp(k, "-v")) v = atof(val);
    else if (!strcmp(k, "-w")) w = atof(val);
    else if (!strcmp(k, "-alpha")) alpha = atof(val);
    else if (!strcmp(k, "-ksp_max_it")) max_it = atoi(val);
    else if (!strcmp(k, "-ksp_rtol")) rtol = atof(val);
    else if (!strcmp(k, "-host_only")) host_only = atoi(val);
    else if (!strcmp(k, "-same_device")) same_device = atoi(val);
    else if (!strcmp(k, "-layout_hint")) use_hint = atoi(val);
    else if (!strcmp(k, "-transport")) transport = !strcmp(val, "rccl") ? AIRHIP_MPI_TRANSPORT_RCCL : AIRHIP_MPI_TRANSPORT_STAGED;
    else if (!strncmp(k, "-pc_air_", 8) && nopts < 64) { names[nopts] = k + 8; values[nopts] = val; nopts++; }
    else DIE("unknown option %s", k);
  }
  if (gy < 0) gy = gx;
  if (gz < 0) gz = dim == 3 ? gx : 1;
  if (!host_only) {
    int ndev = 0;
    if (airhip_device_count(&ndev) != AIRHIP_OK || ndev == 0) DIE("no HIP device (use -host_only 1 to stop after the distribution)");
    if (airhip_set_device(same_device ? 0 : rank % ndev) != AIRHIP_OK) DIE("airhip_set_device: %s", airhip_last_error());
  }
  /* the operator: every rank generates it and keeps its own rows (a test driver; an application assembles its rows) */
  airhip_hostmat A; airhip_csr G;
  if (airhip_gen_adv_diff_fd(dim, gx, gy, gz, u, v, w, 1, alpha, 1.0, 1.0, 1.0, 1, &A) != AIRHIP_OK) DIE("generator: %s", airhip_last_error());
  if (airhip_hostmat_get(A, &G) != AIRHIP_OK) DIE("hostmat_get: %s", airhip_last_error());
  const int64_t n = G.m, r0 = n * rank / size, r1 = n * (rank + 1) / size;
  airhip_csr L;
  L.m = r1 - r0; L.n = n; L.rowptr = G.rowptr + r0; L.col = G.col; L.val = G.val;   /* rowptr stays absolute: the layer uses differences */
  /* col / val are addressed through rowptr[0]: hand over pointers that rowptr indexes correctly */
  /* what a DM would tell the preconditioner: the 8^3 brick of each of this rank's grid points (device layout only) */
  int64_t* hint = (int64_t*)malloc(sizeof(int64_t) * (size_t)(L.m ? L.m : 1));
  {
    const int64_t B = 8, nbx = (gx + B - 1) / B, nby = (gy + B - 1) / B;
    for (int64_t r = r0; r < r1; r++) {
      const int64_t i = r % gx, j = (r / gx) % gy, k = r / (gx * gy);
      hint[r - r0] = (((k / B) * nby + (j / B)) * nbx + (i / B)) * (B * B * B) + ((k % B) * B + (j % B)) * B + (i % B);
    }
  }
  airhip_mpi_pcair pc = NULL;
  double st[5];
  if (airhip_mpi_pcair_setup(MPI_COMM_WORLD, &L, nopts, names, values, use_hint ? hint : NULL, transport, host_only, &pc, st) != AIRHIP_OK)
    DIE("rank %d: setup failed: %s", rank, airhip_mpi_last_error());
  if (host_only) {
    double piece = st[2], total = 0.0;
    MPI_Reduce(&piece, &total, 1, MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
    if (rank == 0) printf("adv_diff_fd_mpi ranks=%d n=%lld levels=%d pieces=%.0f bytes (host only)\n", size, (long long)n, (int)st[0], total);
    airhip_mpi_pcair_destroy(&pc);
    airhip_hostmat_destroy(&A);
    free(hint);
    MPI_Finalize();
    return 0;
  }
  const int64_t m = L.m;
  double *b, *x, *hx = (double*)malloc(sizeof(double) * (size_t)(m ? m : 1));
  for (int64_t i = 0; i < m; i++) hx[i] = 1.0;                 /* x0 = 1, b = 0 (tests/adv_diff_fd.c:179,325) */
  if (hipMalloc((void**)&b, sizeof(double) * (size_t)(m ? m : 1)) != hipSuccess || hipMalloc((void**)&x, sizeof(double) * (size_t)(m ? m : 1)) != hipSuccess) DIE("hipMalloc");
  hipMemset(b, 0, sizeof(double) * (size_t)m);
  hipMemcpy(x, hx, sizeof(double) * (size_t)m, hipMemcpyHostToDevice);
  airhip_ksp_options o; airhip_ksp_result res;
  airhip_ksp_default_options(&o);
  o.type = 1; o.pc_side = 1; o.rtol = rtol; o.atol = 1e-50; o.max_it = max_it;
  if (airhip_mpi_ksp_solve(pc, &o, b, x, &res) != AIRHIP_OK) DIE("rank %d: solve failed: %s", rank, airhip_mpi_last_error());
  hipMemcpy(hx, x, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost);
  double mx = 0.0, gmx = 0.0;
  for (int64_t i = 0; i < m; i++) { const double a = hx[i] < 0 ? -hx[i] : hx[i]; if (a > mx) mx = a; }
  MPI_Reduce(&mx, &gmx, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
  int64_t ex = 0, msgs = 0, bytes = 0, ars = 0;
  airhip_mpi_pcair_traffic(pc, &ex, &msgs, &bytes, &ars);
  if (rank == 0)
    printf("adv_diff_fd_mpi ranks=%d n=%lld levels=%d iterations=%d reason=%d rnorm=%.6e max|x|=%.3e exchanges=%lld messages=%lld allreduces=%lld\n",
           size, (long long)n, (int)st[0], res.its, res.reason, res.rnorm, gmx, (long long)ex, (long long)msgs, (long long)ars);
  hipFree(b); hipFree(x); free(hx); free(hint);
  airhip_mpi_pcair_destroy(&pc);
  airhip_hostmat_destroy(&A);
  MPI_Finalize();
  return res.reason > 0 ? 0 : 1;
}
