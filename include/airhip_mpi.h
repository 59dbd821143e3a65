/* airhip_mpi.h - MPI host layer over the C ABI of include/airhip.h: PCAIR on the ranks of an MPI communicator,
 * one rank per GPU, rows distributed in contiguous blocks (PETSc's MPIAIJ ownership).
 *
 * What it replaces in the reference: the MPIAIJ branch of PCSetUp_AIR_c / PCApply_AIR_c (src/PCAIR.c:150-201) as it
 * runs under `mpiexec -n N` - every rank holds its rows (Ad | Ao with a column map, src/PMISR_Module.F90:174-180),
 * coarse rows stay with their parent C point (src/Grid_Transfer.F90:745-746), off-process columns are fetched in
 * front of every product.  pflare_amd/csrc/petsc_glue.c calls airhip_mpi_pcair_setup for a communicator of size > 1
 * after turning the Mat into `local` with MatMPIAIJGetSeqAIJ; examples/adv_diff_fd_mpi.c calls it directly.
 *
 * Built as pflare_amd/libairhip_mpi.so (make -C pflare_amd/csrc mpi MPI_INC=... MPI_LIB=...), separate from
 * libairhip.so so that the core library has no MPI dependency.
 */
#ifndef AIRHIP_MPI_H
#define AIRHIP_MPI_H
#include <mpi.h>
#include "airhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* halo transport of the distributed cycle */
#define AIRHIP_MPI_TRANSPORT_STAGED 0 /* device -> pinned host -> MPI_Isend / MPI_Irecv -> device; works with several ranks per GPU */
#define AIRHIP_MPI_TRANSPORT_RCCL 1   /* RCCL send / recv groups on the library stream over xGMI (one rank per GPU)             */

typedef struct airhip_mpi_pcair_s* airhip_mpi_pcair;

const char* airhip_mpi_last_error(void);

/* Collective over comm.
 *   local   this rank's rows of the operator: m = owned rows, n = GLOBAL number of columns, GLOBAL column indices
 *           ascending within a row; rank r owns the rows that follow rank r-1's (MPIAIJ ownership).  rowptr[0] need not
 *           be 0: row i holds col / val [rowptr[i], rowptr[i+1]) (a window into a larger CSR is fine)
 *   names / values   nopts options as for airhip_pcair_set_option ("-pc_air_" prefix removed); rank 0's are used
 *   hint    optional locality key per owned row for the device layout (airhip_pcair_set_ordering_hint), or NULL
 *   host_only != 0: stop after every rank has received its piece of the hierarchy (no device needed; CPU tests) -
 *           apply / solve then fail with AIRHIP_ERR_STATE
 * Rank 0 gathers the operator, builds the hierarchy (bit-reproducible: the same for every communicator size), cuts it
 * with airhip_partition_* and ships every rank its piece; every rank uploads its piece.  stats (may be NULL) receives
 * {levels, global rows, bytes of this rank's piece, seconds of setup on rank 0, seconds of cut + ship + upload}. */
int airhip_mpi_pcair_setup(MPI_Comm comm, const airhip_csr* local, int nopts, const char* const* names,
                           const char* const* values, const int64_t* hint, int transport, int host_only,
                           airhip_mpi_pcair* out, double* stats5);
/* y = M x on this rank's rows (device pointers, local row order); collective */
int airhip_mpi_pcair_apply(airhip_mpi_pcair pc, const double* d_x, double* d_y);
/* right- or left-preconditioned Krylov solve of A x = b with the distributed operator and PCAIR; collective */
int airhip_mpi_ksp_solve(airhip_mpi_pcair pc, const airhip_ksp_options* o, const double* d_b, double* d_x,
                         airhip_ksp_result* res);
/* the rank-local PCAIR object (borrowed) and communicator: everything in airhip.h that takes them works on them */
int airhip_mpi_pcair_handle(airhip_mpi_pcair pc, airhip_pcair* inner, airhip_comm* comm);
/* halo traffic counters of this rank since setup: exchanges, messages, bytes, all-reduces */
int airhip_mpi_pcair_traffic(airhip_mpi_pcair pc, int64_t* exchanges, int64_t* messages, int64_t* bytes, int64_t* allreduces);
int airhip_mpi_pcair_destroy(airhip_mpi_pcair* pc);

#ifdef __cplusplus
}
#endif
#endif
