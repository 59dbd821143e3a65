/* airhip.h - C ABI of the MI355X-native PCAIR / PCPFLAREINV solve phase.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A PETSc
 * plugin or any C / Fortran (bind(C)) / ctypes caller binds exactly
 * these symbols.  Every function returns 0 on success (PETSC_SUCCESS convention) and a
 * negative AIRHIP_ERR_* code otherwise; airhip_last_error() returns the message.  Library
 * code never aborts the process (the reference's Fortran callbacks MPI_Abort,
 * src/FC_Smooth.F90:449-452).
 *
 * Reference interfaces each entry point replaces (paths in PFLAREProject/PFLARE):
 *   airhip_pcair_create        PCCreate_AIR                 src/PCAIR.c:3601-3644
 *   airhip_pcair_set_option    PCSetFromOptions_AIR_c /     src/PCAIR.c:2777-3033,
 *                              PCAIRSet* setters            include/pflare.h:121-243
 *   airhip_pcair_setup         PCSetUp_AIR_c -> setup_air_pcmg   src/PCAIR.c:182-201,
 *                                                           src/AIR_MG_Setup.F90:44
 *   airhip_pcair_set_level /   the per-level state of air_multigrid_data filled by the
 *   airhip_pcair_set_coarse /  reference's own setup         src/AIR_Data_Type.F90:284-360
 *   airhip_pcair_finalize
 *   airhip_pcair_apply         PCApply_AIR_c -> PCApply_AIR_Shell -> PCMG Kaskade cycle ->
 *                              mg_FC_point_richardson / mg_coarse_shell_apply
 *                              src/PCAIR.c:150-166, src/PCAIR_Shell.F90:170-188,
 *                              src/FC_Smooth.F90:29-49,421-640
 *   airhip_pcair_destroy       PCDestroy_AIR_c / destroy_air_data  src/PCAIR_Shell.F90:192-211
 *   airhip_pcinv_*             PCCreate_PFLAREINV / PCSetUp_PFLAREINV_c / PCApply_PFLAREINV_c
 *                              src/PCPFLAREINV.c:887-951, 689-783, 618-626
 *   airhip_ksp_solve           KSPSolve (PETSc KSPRICHARDSON / KSPGMRES), the caller of PCApply
 *   airhip_mat_*               MatMult on a device-resident AIJ matrix (PETSc MatMult_SeqAIJ*)
 *
 * Vectors are raw device pointers (HIP) to fp64 data owned by the caller.  Matrices are
 * handed over as host CSR (int64 row pointers, int32 column indices sorted ascending, fp64
 * values - what MatSeqAIJGetCSR / MatGetRowIJ give) and copied to the device at set time.
 */
#ifndef AIRHIP_H
#define AIRHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIRHIP_OK 0
#define AIRHIP_ERR_ARG (-1)
#define AIRHIP_ERR_STATE (-2)
#define AIRHIP_ERR_HIP (-3)
#define AIRHIP_ERR_UNSUPPORTED (-4)
#define AIRHIP_ERR_INTERNAL (-5)

/* PCPFLAREINVType: same enumerators and values as the reference's enum (include/pflare.h:36-46), so that a C
 * caller can pass a PCPFLAREINVType straight through.  tests/test_capi.py parses both headers and compares them.
 * The options themselves travel as strings (airhip_pcair_set_option "inverse_type" "newton" ...). */
#define AIRHIP_INV_POWER 0
#define AIRHIP_INV_ARNOLDI 1
#define AIRHIP_INV_NEWTON 2
#define AIRHIP_INV_NEWTON_NO_EXTRA 3
#define AIRHIP_INV_NEUMANN 4
#define AIRHIP_INV_SAI 5
#define AIRHIP_INV_ISAI 6
#define AIRHIP_INV_WJACOBI 7
#define AIRHIP_INV_JACOBI 8

/* kinds of approximate inverse as applied in the solve phase */
#define AIRHIP_OP_NONE 0
#define AIRHIP_OP_ASSEMBLED 1 /* CSR                                   */
#define AIRHIP_OP_DIAG 2      /* vector (MATDIAGONAL)                  */
#define AIRHIP_OP_HORNER 3    /* monomial coefficients, matrix-free    */
#define AIRHIP_OP_NEWTON 4    /* Newton-basis roots, matrix-free       */

typedef struct {
  int64_t m, n;
  const int64_t* rowptr; /* m+1 */
  const int32_t* col;    /* nnz, ascending within a row */
  const double* val;     /* nnz */
} airhip_csr;

/* An approximate inverse (assembled Mat or MatShell + mat_ctxtype in the reference). */
typedef struct {
  int kind;              /* AIRHIP_OP_*                                               */
  airhip_csr mat;        /* ASSEMBLED                                                 */
  const double* diag;    /* DIAG: n entries                                           */
  int ncoef;             /* HORNER: coefficients[ncoef] constant term first;          */
  const double* coef;    /*         NEWTON: ncoef roots, real parts                   */
  const double* coef_im; /* NEWTON: imaginary parts                                   */
  int diag_scale;        /* q(D^-1 A) D^-1                                            */
  int neumann;           /* operator is I - D^-1 A (Neumann series)                   */
} airhip_invop;

typedef struct airhip_pcair_s* airhip_pcair;
typedef struct airhip_pcinv_s* airhip_pcinv;
typedef struct airhip_mat_s* airhip_mat;

const char* airhip_last_error(void);
int airhip_version(void);
/* number of HIP devices visible / select the device used by subsequently created objects */
int airhip_device_count(int* count);
int airhip_set_device(int device);
/* host threads used by the setup phase (OpenMP); returns the previous setting through *prev
 * (may be null).  n <= 0 leaves the setting unchanged. */
int airhip_set_num_threads(int n, int* prev);

/* ------------------------------------------------------------- tuning and inspection hooks
 * Declared so that nothing the library exports is hidden; NOT part of the stable ABI.
 * airhip_set_tuning: process-wide knobs for experiments and tests.  Keys: "spgemm_pool_entries" (caps the output pool
 *   of the device SpGEMM at that many entries, 0 = sized from the operands; a small cap forces the multi-round path),
 *   "spgemm_pool_chunk" (entries a wave takes from that pool at a time, -1 = automatic), "block_order_min_blocks"
 *   (operators with at least this many row blocks are swept in the spatial order of their rows, default 4096), "fc_fuse"
 *   (default 1; 0 runs the one-point prolongation and the A_fc product of an F relaxation as two launches - diagnostic),
 *   "mb16" / "mb16_max_rows" (default 1 / 128: row blocks of at most that many rows are stored with 16-bit multi-base index
 *   words - 10 instead of 12 bytes per non-zero; 0 = the packed / raw 32-bit forms only; act at the next conversion),
 *   "vdict" (default 1: a multi-base block whose values take at most 256 distinct bit patterns stores them once and one-byte
 *   indices - the operators of the first levels of a constant-coefficient problem: 8 nd + 512 bytes of values per block
 *   instead of 8 per non-zero; lossless, the same doubles are multiplied in the same order; 0 = plain values; acts at the next
 *   conversion),
 *   "occ_spmv" / "occ_fc" (default 6 / 6: workgroups per CU = waves per SIMD of the SpMV / fused F-relaxation kernels, 0 =
 *   what the registers allow; act at the next launch), "fc_spatial_x" (default 1: A_fc gathers from a copy of x_c in the
 *   spatial order of the C points on the levels where that pays, 0 = never, 2 = every level; acts at the next upload),
 *   "krylov_blocks" (default 512: workgroups of the multi-dot / multi-axpy launches of the Krylov solvers),
 *   "ksp_unnormalised" (default 1: GMRES keeps its basis vectors unnormalised and their norms on the host - no scaling
 *   pass per iteration; 0: normalised basis.  Same Hessenberg matrix up to rounding, identical iteration counts),
 *   "fuse_first_sweep" (default 1, acts at the next upload: on the levels whose relaxation starts with F sweeps - one-point W,
 *   sparse A_ff, A_ff inverse assembled or a matrix-free polynomial - the first sweep runs on G = A_fc + A_ff W, four products
 *   per level instead of the reference's five (matrix-free: one product with A_ff less); same iterates up to rounding (1e-13 per cycle), NOT bit for bit; 0 = the reference's sequence; 2 = as 1
 *   and A_fc stays on the device, so that "fuse_first_sweep_run" (default 1) can switch between the two sequences on one
 *   hierarchy), "profile_chain" (default 1: inside the cycle the stop event of one profiled kernel is the start event of the
 *   next; 0 = a pair of events per kernel),
 *   "perm_chunks" (default 1: the natural <-> hierarchical permutation of PCApply / KSPSolve vectors walks the key range chunk
 *   by chunk - every section's share of a chunk in one workgroup - instead of one plain gather; 0 = plain gathers; bitwise
 *   neutral),
 *   "ksp_keep_preconditioned" (default 1: right-preconditioned GMRES keeps z_j = M v_j, restart more vectors, and updates
 *   x += Z y without the extra PCApply of PETSc's x += M (V y); same Arnoldi process, Hessenberg matrix, residual norms and
 *   iteration count bit for bit - the solution differs in its last rounding; 0: PETSc's form).
 *   Bitwise neutral - not a bit of any result changes - are the FORMAT, OCCUPANCY and ORDER keys ("mb16", "mb16_max_rows", "vdict",
 *   "occ_spmv", "occ_fc", "fc_spatial_x", "block_order_min_blocks", "fc_fuse", the "spgemm_*" keys).  NOT bitwise neutral:
 *   "ksp_unnormalised" (every Hessenberg entry is rounded differently), "krylov_blocks" (the grid sets the summation order
 *   of every dot product and norm), "ksp_keep_preconditioned" (the last rounding of x) and "fuse_first_sweep" (the sums of the
 *   first F sweep are re-associated); they keep the iteration counts and agree within the tolerances
 *   tests/test_gpu_ksp_variants.py and tests/test_fused_first_sweep.py state.
 * airhip_pcair_setup_host_only / airhip_pcinv_setup_host_only: build the hierarchy without touching a device (CPU-side
 *   tests of the setup, hierarchy export); apply / solve then fail with AIRHIP_ERR_STATE.
 * airhip_spgemm_filtered: C = filter(A*B) with the host (on_device = 0) or the device SpGEMM of the setup, drop / lump /
 *   negate as in the setup's Z and RAP products; stats[5] = rows each accumulator kind worked on (registers, LDS tables
 *   of 1k / 2k / 4k slots, global-memory table). */
int airhip_set_tuning(const char* key, int value);

/* ---------------------------------------------------------------- stand-alone CF splitting */
/* compute_cf_splitting(Mat, int symmetric, PetscReal strong_threshold, int max_luby_steps, int cf_splitting_type,
 * int ddc_its, PetscReal fraction_swap, IS* is_fine, IS* is_coarse)  (include/pflare.h:108-113, src/CF_Splitting.F90:235-392).
 * cf_splitting_type: AIRHIP_CF_PMISR_DDC | AIRHIP_CF_DIAG_DOM | AIRHIP_CF_CR (the reference's enum values); for
 * DIAG_DOM / CR strong_threshold carries the target ratio / rate.  CR relaxes with the default A_ff inverse
 * (Arnoldi basis, order 6, sparsity 1).  markers[i] = -1 (F point) or +1 (C point), i < A->m; host arrays. */
#define AIRHIP_CF_PMISR_DDC 0
#define AIRHIP_CF_DIAG_DOM 1
#define AIRHIP_CF_CR 6
int airhip_compute_cf_splitting(const airhip_csr* A, int symmetric, double strong_threshold, int max_luby_steps,
                                int cf_splitting_type, int ddc_its, double fraction_swap, int32_t* markers);
/* the same splitting (PMISR + DDC only) computed by the device kernels; markers are identical to the host's
 * (reference device twins: src/PMISR_Modulek.kokkos.cxx, src/DDC_Modulek.kokkos.cxx, src/MatDiagDomk.kokkos.cxx) */
int airhip_compute_cf_splitting_on_device(const airhip_csr* A, int symmetric, double strong_threshold, int max_luby_steps,
                                          int ddc_its, double fraction_swap, int32_t* markers);

/* ---------------------------------------------------------------- synthetic operators */
/* tests/adv_1d.c:80-105 and tests/adv_diff_fd.c:366-586.  The CSR arrays are owned by the
 * returned handle; release with airhip_hostmat_destroy. */
typedef struct airhip_hostmat_s* airhip_hostmat;
int airhip_gen_adv_1d(int64_t n, airhip_hostmat* out);
int airhip_gen_adv_diff_fd(int dim, int64_t M, int64_t N, int64_t P, double u, double v, double w,
                           int unit_velocity, double alpha, double L_x, double L_y, double L_z,
                           int adv_nondim, airhip_hostmat* out);
int airhip_hostmat_from_csr(const airhip_csr* A, airhip_hostmat* out); /* deep copy */
int airhip_hostmat_get(airhip_hostmat A, airhip_csr* view);
int airhip_hostmat_destroy(airhip_hostmat* A);
/* compute_diag_dom_submatrix(Mat, PetscReal max_dd_ratio, Mat* output)  (include/pflare.h:114, src/MatDiagDomSubmatrix.F90:17-76):
 * A(F, F) for the diag_dom splitting with strong_threshold = max_dd_ratio, 0 < max_dd_ratio < 1; release with
 * airhip_hostmat_destroy */
int airhip_compute_diag_dom_submatrix(const airhip_csr* A, double max_dd_ratio, airhip_hostmat* out);

/* ---------------------------------------------------------------------------- PCAIR */
int airhip_pcair_create(airhip_pcair* pc);
/* name without the "-pc_air_" prefix, value as it would appear on a PETSc command line,
 * e.g. ("a_drop","1e-3"), ("inverse_type","power"), ("smooth_type","ff"), ("a_lump","1").
 * Also accepted: ("mg_coarse_ksp_max_it","10") for -mg_coarse_ksp_type richardson. */
int airhip_pcair_set_option(airhip_pcair pc, const char* name, const char* value);
/* Optional locality hint for the DEVICE LAYOUT: key[i] for every row i of the operator (e.g. a brick
 * or Morton index of the grid point a DM would know).  Rows with close keys are stored close together
 * on every level (coarse rows inherit the key of their parent C point), which shortens the reach of the
 * x gathers.  The hierarchy, the numbering of the caller's vectors and the results are unaffected.
 * Call before setup / finalize / partition_create. */
int airhip_pcair_set_ordering_hint(airhip_pcair pc, const int64_t* key, int64_t n);
/* Full setup from the operator: host hierarchy construction + device upload. */
int airhip_pcair_setup(airhip_pcair pc, const airhip_csr* A);
int airhip_pcair_setup_host_only(airhip_pcair pc, const airhip_csr* A);   /* see "tuning and inspection hooks" */

/* Bring-your-own hierarchy (e.g. built by PFLARE's own setup under PETSc).  Levels are
 * numbered 0 (finest) .. no_levels-2, all index lists and matrices in the reference's
 * level-local numbering: F,C ascending row indices of the level operator; A_ff (n_f x n_f)
 * or aff_diag (n_f, when A_ff is MATDIAGONAL); A_fc (n_f x n_c); Z (n_c x n_f, the Z block
 * of R=[Z I]); W either one-point (wcol[n_f], -1 = empty row) or general CSR (n_f x n_c);
 * optional A_cf, A_cc, inv_cc for C smoothing; smooth_order as in air_options. */
int airhip_pcair_begin_levels(airhip_pcair pc, int no_levels, const airhip_csr* A_top);
int airhip_pcair_set_level(airhip_pcair pc, int level, int64_t n, int64_t n_f, int64_t n_c,
                           const int32_t* F, const int32_t* C,
                           const airhip_csr* A_ff, const double* aff_diag,
                           const airhip_csr* A_fc, const airhip_invop* inv_ff,
                           const airhip_csr* Z, const int32_t* wcol, const airhip_csr* W,
                           const airhip_csr* A_cf, const airhip_csr* A_cc, const airhip_invop* inv_cc,
                           const int* smooth_order10);
int airhip_pcair_set_coarse(airhip_pcair pc, const airhip_csr* A_coarse, const airhip_invop* inv,
                            int coarse_richardson_its);
int airhip_pcair_finalize(airhip_pcair pc);

/* Stream contract of every call that reads or writes caller vectors (apply, mult, solve).
 * The library launches on a private non-blocking stream per object.  On entry that stream is made to wait (event, no
 * host synchronisation) for everything already queued on the CALLER'S stream, so vectors produced by earlier
 * kernels of the caller are complete before they are read.  The caller's stream is the `stream` argument of the
 * *_on_stream entry points (a hipStream_t passed as void*; NULL = the legacy default stream) and the legacy default
 * stream for the entry points without one.
 *   - *_on_stream variants of apply / mult also make the caller's stream wait for the result on exit: they return
 *     without synchronising the host and the output may be consumed by work queued on `stream` afterwards.  This is
 *     what a host Krylov solver (PETSc KSP_PCApply) needs: no host synchronisation per PCApply
 *     (the reference synchronises after every gather / scatter: src/VecISCopyLocalk.kokkos.cxx:192).
 *   - airhip_pcair_apply / airhip_pcinv_apply return after queueing; the result is complete after
 *     airhip_pcair_sync / airhip_pcinv_sync (or a later solve on the same object).
 *   - airhip_mat_mult and airhip_ksp_solve synchronise the host before they return. */

/* y = M x : one AIR V-cycle, natural ordering, device pointers (PCApply_AIR_c, src/PCAIR.c:150-166). */
int airhip_pcair_apply(airhip_pcair pc, const double* d_x, double* d_y);
int airhip_pcair_apply_on_stream(airhip_pcair pc, const double* d_x, double* d_y, void* stream);
int airhip_pcair_sync(airhip_pcair pc);
int airhip_pcair_destroy(airhip_pcair* pc);

/* hierarchy inspection / export (host arrays valid until the next setup / destroy) */
int airhip_pcair_num_levels(airhip_pcair pc, int* no_levels);
int airhip_pcair_level_sizes(airhip_pcair pc, int level, int64_t* n, int64_t* n_f, int64_t* n_c,
                             int* aff_diag, int* one_point, int64_t* nnz_A);
/* which: "F","C" (int32 index lists via idx/len), "A_ff","A_fc","A_cf","A_cc","A_full","Z","W","M_ff",
 * "M_cc" (CSR), "aff_diag","M_ff_diag","M_cc_diag" (vector via vec/len), "wcol" (idx). */
int airhip_pcair_level_get(airhip_pcair pc, int level, const char* which, airhip_csr* csr,
                           const int32_t** idx, const double** vec, int64_t* len);
int airhip_pcair_level_inv(airhip_pcair pc, int level, int which_c, airhip_invop* inv); /* level == no_levels-1: coarse */
int airhip_pcair_level_smooth_order(airhip_pcair pc, int level, int* smooth_order10);
int airhip_pcair_coarse_get(airhip_pcair pc, airhip_csr* A_coarse);
/* complexities as printed by -pc_air_print_stats_timings (src/AIR_MG_Stats.F90:79-382) and
 * the algorithmic bytes of one V-cycle */
int airhip_pcair_complexities(airhip_pcair pc, double* grid, double* op, double* cycle, double* storage,
                              double* cycle_alg_bytes);
/* algorithmic bytes of one V-cycle on this device (this rank's share when distributed) */
int airhip_pcair_cycle_alg_bytes(airhip_pcair pc, double* bytes);
int airhip_pcair_setup_times(airhip_pcair pc, double* host_setup_s, double* upload_s);
/* free the host copy of the hierarchy (export functions then fail with AIRHIP_ERR_STATE) */
/* A hierarchy set up on the device keeps the operators no host routine reads (A_cf, the assembled A_ff inverses, Z) there
 * during the setup; a thread of the library copies them into the host hierarchy after airhip_pcair_setup has returned (the
 * solve does not need them).  Every call that reads the host hierarchy (level_get, level_inv, coarse_get, partition_create,
 * release_host, a new setup, destroy) waits for that thread by itself; this call only waits, and reports how long the
 * thread ran.  AIRHIP_DEFER_DOWNLOADS=0 in the environment: everything is copied during the setup, as in the reference's
 * host build (src/AIR_MG_Setup.F90 keeps all operators as host Mats). */
int airhip_pcair_host_mirror_wait(airhip_pcair pc, double* seconds);
int airhip_pcair_release_host(airhip_pcair pc);
/* run the V-cycle through a captured HIP graph (default off) */
int airhip_pcair_use_graph(airhip_pcair pc, int on);

/* per-kernel HIP-event timing of the V-cycle kernels (on the launch stream) */
int airhip_pcair_profile(airhip_pcair pc, int on);
int airhip_pcair_profile_reset(airhip_pcair pc);
int airhip_pcair_profile_count(airhip_pcair pc, int* n);
int airhip_pcair_profile_get(airhip_pcair pc, int i, const char** name, int* level, double* alg_bytes,
                             int64_t* calls, double* total_ms);
/* name of the HIP kernel function record i launches (the name a rocprofv3 kernel trace shows) */
int airhip_pcair_profile_kernel(airhip_pcair pc, int i, const char** kernel);

/* ---------------------------------------------------------------------- multi-GPU (8e)
 * Row-block distribution, one process per GPU - the reference's MPIAIJ model (rank-owned rows,
 * off-process columns fetched before each product; src/PMISR_Module.F90:174-180,
 * src/Grid_Transfer.F90:656-668; coarse rows stay with their parent C point, :745-746).
 * One rank builds the hierarchy (option "device_upload" = "0" keeps it on the host), cuts it
 * with airhip_partition_*, and ships each rank its piece as an archive of named flat arrays
 * (the transport is the caller's: torch.distributed in bench.py).  Every rank then calls
 * airhip_pcair_load_distributed; apply / ksp_solve work on the rank's rows (natural order of
 * its owned rows, ascending global index) and are collective. */
typedef struct airhip_partition_s* airhip_partition;
typedef struct airhip_archive_s* airhip_archive;
typedef struct airhip_comm_s* airhip_comm;
#define AIRHIP_I32 0
#define AIRHIP_I64 1
#define AIRHIP_F64 2
/* owner[i] in [0,nranks): the rank owning fine row i.  A_top must stay valid until destroy. */
int airhip_partition_create(airhip_pcair pc, const airhip_csr* A_top, int nranks, const int32_t* owner,
                            airhip_partition* out);
int airhip_partition_extract(airhip_partition p, int rank, airhip_archive* piece);
int airhip_partition_destroy(airhip_partition* p);
int airhip_archive_create(airhip_archive* out);
/* adds a view of caller memory (must outlive the archive's use) */
int airhip_archive_add(airhip_archive ar, const char* name, int dtype, const void* data, int64_t count);
int airhip_archive_count(airhip_archive ar, int* n);
int airhip_archive_item(airhip_archive ar, int i, const char** name, int* dtype, const void** data, int64_t* count);
int airhip_archive_destroy(airhip_archive* ar);
/* transports of the halo exchange: RCCL (device buffers, xGMI; unique id from rank 0, broadcast
 * by the caller) or host-staged callbacks (MPI in a host application; several ranks per GPU). */
typedef int (*airhip_exchange_fn)(void* user, int npeers, const int* peers, const double* send,
                                  const int64_t* send_off, double* recv, const int64_t* recv_off);
typedef int (*airhip_allreduce_fn)(void* user, double* vals, int n); /* in-place sum over ranks */
int airhip_comm_rccl_unique_id(void* out128);
int airhip_comm_create_rccl(int rank, int size, const void* unique_id128, airhip_comm* out);
int airhip_comm_create_staged(int rank, int size, airhip_exchange_fn ex, airhip_allreduce_fn ar, void* user,
                              airhip_comm* out);
int airhip_comm_stats(airhip_comm c, int64_t* exchanges, int64_t* allreduces);
/* halo traffic this rank has sent since the communicator was created (or its last self-test): bytes, and messages
 * (one per peer with something to send per exchange) */
int airhip_comm_traffic(airhip_comm c, int64_t* bytes_sent, int64_t* messages_sent);
/* collective check of a transport before it is trusted with a solve: every rank sends `count` doubles of a
 * known pattern to every other rank (device buffers, twice) and sums a value over the ranks; *ok = 1 if
 * everything arrived intact on this rank */
int airhip_comm_selftest(airhip_comm c, int64_t count, int* ok);
int airhip_comm_destroy(airhip_comm* c);
/* the communicator must outlive the PCAIR object */
int airhip_pcair_load_distributed(airhip_pcair pc, airhip_archive piece, airhip_comm comm);

/* ------------------------------------------------------------------------ PCPFLAREINV */
int airhip_pcinv_create(airhip_pcinv* pc);
/* names: "type" (power|arnoldi|newton|neumann|wjacobi|jacobi), "poly_order", "sparsity_order",
 * "matrix_free", "diag_scale" */
int airhip_pcinv_set_option(airhip_pcinv pc, const char* name, const char* value);
int airhip_pcinv_setup(airhip_pcinv pc, const airhip_csr* A);
int airhip_pcinv_setup_host_only(airhip_pcinv pc, const airhip_csr* A);
int airhip_spgemm_filtered(const airhip_csr* A, const airhip_csr* B, int drop, double tol, int lump, int negate,
                           int on_device, airhip_hostmat* out, int64_t* stats);
/* inspection hook, not part of the stable ABI: G = A_fc + A_ff W for the one-point prolongator W (wcol[i] = W's column of F
 * point i, -1 = none) - the operator the fused first F sweep of the cycle runs on (airhip_set_tuning "fuse_first_sweep"),
 * built by the host code (on_device = 0) or by the device code of the upload; both give the same bits */
int airhip_fused_first_operator(const airhip_csr* A_ff, const airhip_csr* A_fc, const int32_t* wcol, int on_device, airhip_hostmat* out);
int airhip_pcinv_apply(airhip_pcinv pc, const double* d_x, double* d_y);      /* PCApply_PFLAREINV_c, src/PCPFLAREINV.c:618-626 */
int airhip_pcinv_apply_on_stream(airhip_pcinv pc, const double* d_x, double* d_y, void* stream);
/* the zero-level PCAIR object a PCPFLAREINV is built on - BORROWED, valid while pc is; what airhip_partition_create takes
 * to cut a PCPFLAREINV for a multi-GPU run (the per-rank objects are then PCAIR objects loaded from the pieces) */
int airhip_pcinv_as_pcair(airhip_pcinv pc, airhip_pcair* out);
int airhip_pcinv_sync(airhip_pcinv pc);
int airhip_pcinv_get_inv(airhip_pcinv pc, airhip_invop* inv);
int airhip_pcinv_destroy(airhip_pcinv* pc);

/* ------------------------------------------------------------------ device matrices */
int airhip_mat_create(const airhip_csr* A, airhip_mat* out);
int airhip_mat_mult(airhip_mat A, const double* d_x, double* d_y); /* y = A x, synchronous */
int airhip_mat_mult_on_stream(airhip_mat A, const double* d_x, double* d_y, void* stream); /* asynchronous, see above */
/* diagnostic (the format tests decode it and compare with the CSR that went in, entry for entry): the row-block form of the
 * matrix as it lies on the device.  Sizes first (desc == NULL), then the arrays: desc = nblocks descriptors of 32 bytes
 * {int64 first value, int32 first row, int32 kind / base, int32 non-zeros, int32 rows, int64 byte offset of the index region},
 * idx = the index regions (idx_bytes), val = the value regions back to back - *nnz returns their length in doubles: the number
 * of non-zeros unless blocks hold value dictionaries (table + 64 doubles of one-byte indices instead of the block's values);
 * roff = one 16-bit offset per row (csrc/device/dev.hpp). */
int airhip_mat_blocks(airhip_mat A, int64_t* nblocks, int64_t* idx_bytes, int64_t* nnz, void* desc, void* idx, double* val, void* roff);
int airhip_mat_destroy(airhip_mat* A);

/* ---------------------------------------------------------------------------- KSP */
typedef struct {
  int type;      /* 0 richardson, 1 gmres                                   */
  int pc_side;   /* 0 left, 1 right (gmres)                                  */
  int norm_type; /* richardson: 0 preconditioned, 1 unpreconditioned         */
  double rtol, atol, dtol;
  int max_it;
  int restart;
} airhip_ksp_options;
typedef struct {
  int its;
  int reason;      /* KSPConvergedReason values: 2 RTOL, 3 ATOL, -3 ITS, -4 DTOL, -5 BREAKDOWN */
  double rnorm;
  double solve_ms; /* HIP-event time of the solve on the library stream                */
  int pc_applies;  /* PCApply calls of the solve.  Right-preconditioned GMRES keeps the preconditioned basis z_j = M v_j
                    * (airhip_set_tuning "ksp_keep_preconditioned", default 1) and updates x += Z y, so this equals `its`;
                    * PETSc's form x += M (V y) (key = 0) applies M once more per restart cycle                          */
  int reserved;
} airhip_ksp_result;
void airhip_ksp_default_options(airhip_ksp_options* o);
/* Solve A x = b (natural ordering, device pointers; x holds the initial guess).  Exactly one
 * of pcair / pcinv may be non-null (both null: unpreconditioned); with pcair the operator is
 * the one given to its setup and `A` may be null.  history (may be null) receives up to
 * history_cap residual norms, *history_len their count. */
int airhip_ksp_solve(const airhip_ksp_options* o, airhip_mat A, airhip_pcair pcair, airhip_pcinv pcinv,
                     const double* d_b, double* d_x, airhip_ksp_result* res,
                     double* history, int history_cap, int* history_len);
/* the same with the caller's stream named (b and x0 may have been produced on it) */
int airhip_ksp_solve_on_stream(const airhip_ksp_options* o, airhip_mat A, airhip_pcair pcair, airhip_pcinv pcinv,
                               const double* d_b, double* d_x, airhip_ksp_result* res,
                               double* history, int history_cap, int* history_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIRHIP_H */
