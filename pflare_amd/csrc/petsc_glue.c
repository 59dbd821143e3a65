/* PETSc plugin glue: PC types "air" and "pflareinv" implemented on libairhip (include/airhip.h).
 *
 * Exports the symbols PETSc / PFLARE users already link against:
 *     PCCreate_AIR, PCRegister_AIR                 (reference: src/PCAIR.c:3601-3652)
 *     PCCreate_PFLAREINV, PCRegister_PFLAREINV     (reference: src/PCPFLAREINV.c:887-957)
 *     PetscDLLibraryRegister_petscpflare           (reference: src/PCAIR.c:3658-3663)
 * with the PC vtable the reference fills (apply, setup, destroy, view, reset, setfromoptions:
 * src/PCAIR.c:3636-3641, src/PCPFLAREINV.c:928-933).
 *
 * Built only when PETSc is present:   make -C pflare_amd/csrc petsc_glue PETSC_DIR=... [PETSC_ARCH=...]
 * -> pflare_amd/libpetscpflare.so, loadable with  -dll_append <path>/libpetscpflare.so  or linked into the
 * application in place of the reference's libpflare.  NOT compiled in the environment this repository was
 * developed in (no PETSc there): every airhip_* call below is exercised through the same ABI by tests/test_capi.py,
 * tests/test_gpu_streams.py and examples/adv_1d.c; tests/test_petsc_glue.py checks this file against the header.
 *
 * Scope: MATSEQAIJ on one rank, and MATMPIAIJ on a communicator of any size when built with -DAIRHIP_WITH_MPI against
 * pflare_amd/libairhip_mpi.so (include/airhip_mpi.h; `make petsc_glue` does that when libairhip_mpi.so exists): every
 * rank hands its rows to airhip_mpi_pcair_setup (MatMPIAIJGetSeqAIJ -> local rows with global columns), rank 0 builds
 * and cuts the hierarchy, every rank uploads its piece; PCApply then works on the rank's rows like the reference's
 * MPIAIJ PCAIR (src/PCAIR.c:182-201, src/AIR_MG_Setup.F90:44-137).  Without AIRHIP_WITH_MPI a communicator of size > 1
 * is refused instead of silently solving a different problem.  PCPFLAREINV is one rank only.
 * Minimum PETSc: 3.20 (PetscDeviceContextGetStreamHandle); the reference itself needs >= 3.25 (its Makefile:10-14).
 * This file has never met a compiler (see above): treat it as reviewed text.
 */
#include <petsc/private/pcimpl.h>
#include <petscmat.h>
#include <petscvec.h>
#include <petscdevice.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "airhip.h"
#if defined(AIRHIP_WITH_MPI)
#include "airhip_mpi.h"
#endif

#define AIRHIP_MAX_OPTS 96

typedef struct {
  int          is_inv;                 /* 0: PCAIR, 1: PCPFLAREINV */
  airhip_pcair air;
  airhip_pcinv inv;
  PetscInt     n;                      /* rows of the operator the object was set up for */
  /* options are kept so that PCReset (which frees the hierarchy) does not forget them */
  int          nopts;
  char         names[AIRHIP_MAX_OPTS][48];
  char         values[AIRHIP_MAX_OPTS][64];
  /* staging for host (non-device) vectors */
  double      *d_x, *d_y;
#if defined(AIRHIP_WITH_MPI)
  airhip_mpi_pcair dist; /* communicator of size > 1: the distributed object; `air` then is its rank-local PCAIR (borrowed) */
#endif
} PC_AIRHIP;

#define AIRHIP_CALL(expr)                                                                              \
  do {                                                                                                 \
    int _rc = (expr);                                                                                  \
    PetscCheck(_rc == AIRHIP_OK, PETSC_COMM_SELF, PETSC_ERR_LIB, "airhip (%d): %s", _rc, airhip_last_error()); \
  } while (0)

/* the -pc_air_* options this library implements (docs/options.md:22-106); anything else the user sets under
 * the prefix is reported by PETSc's -options_left as usual */
static const char *const air_options[] = {
  "max_levels", "coarse_eq_limit", "auto_truncate_start_level", "auto_truncate_tol", "strong_threshold", "ddc_its",
  "ddc_fraction", "max_luby_steps", "cf_splitting_type", "symmetric", "smooth_type", "diag_scale_polys",
  "matrix_free_polys", "one_point_classical_prolong", "full_smoothing_up_and_down", "inverse_type", "poly_order",
  "inverse_sparsity_order", "c_inverse_type", "c_poly_order", "c_inverse_sparsity_order", "coarsest_inverse_type",
  "coarsest_poly_order", "coarsest_inverse_sparsity_order", "coarsest_matrix_free_polys", "coarsest_diag_scale_polys",
  "r_drop", "a_drop", "a_lump", "strong_r_threshold", "z_type", "lair_distance", "improve_z_its", "improve_w_its",
  "print_stats_timings", "processor_agglom", NULL};
static const char *const inv_options[] = {"type", "poly_order", "sparsity_order", "matrix_free", "diag_scale", NULL};

static PetscErrorCode airhip_push_option(PC_AIRHIP *ctx, const char *name, const char *value)
{
  PetscFunctionBegin;
  for (int i = 0; i < ctx->nopts; i++)
    if (!strcmp(ctx->names[i], name)) {
      strncpy(ctx->values[i], value, sizeof(ctx->values[i]) - 1);
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  PetscCheck(ctx->nopts < AIRHIP_MAX_OPTS, PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "too many options");
  strncpy(ctx->names[ctx->nopts], name, sizeof(ctx->names[0]) - 1);
  strncpy(ctx->values[ctx->nopts], value, sizeof(ctx->values[0]) - 1);
  ctx->nopts++;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode airhip_make_handle(PC_AIRHIP *ctx)
{
  PetscFunctionBegin;
  if (ctx->is_inv) {
    if (!ctx->inv) AIRHIP_CALL(airhip_pcinv_create(&ctx->inv));
    for (int i = 0; i < ctx->nopts; i++) AIRHIP_CALL(airhip_pcinv_set_option(ctx->inv, ctx->names[i], ctx->values[i]));
  } else {
    if (!ctx->air) AIRHIP_CALL(airhip_pcair_create(&ctx->air));
    for (int i = 0; i < ctx->nopts; i++) AIRHIP_CALL(airhip_pcair_set_option(ctx->air, ctx->names[i], ctx->values[i]));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetFromOptions_AIR_c (src/PCAIR.c:2777-3033) / PCSetFromOptions_PFLAREINV_c (src/PCPFLAREINV.c:785-883):
 * every option is forwarded verbatim, the library parses names and enum strings itself */
#if PETSC_VERSION_GE(3, 23, 0)
static PetscErrorCode PCSetFromOptions_AIRHIP(PC pc, PetscOptionItems PetscOptionsObject)
#else
static PetscErrorCode PCSetFromOptions_AIRHIP(PC pc, PetscOptionItems *PetscOptionsObject)
#endif
{
  PC_AIRHIP         *ctx    = (PC_AIRHIP *)pc->data;
  const char *const *names  = ctx->is_inv ? inv_options : air_options;
  const char        *family = ctx->is_inv ? "-pc_pflareinv_" : "-pc_air_";
  char               key[96], val[64];
  PetscBool          set;

  PetscFunctionBegin;
  PetscOptionsHeadBegin(PetscOptionsObject, ctx->is_inv ? "PCPFLAREINV options (airhip)" : "PCAIR options (airhip)");
  for (int i = 0; names[i]; i++) {
    PetscCall(PetscSNPrintf(key, sizeof key, "%s%s", family, names[i]));
    PetscCall(PetscOptionsString(key, "forwarded to libairhip", "", "", val, sizeof val, &set));
    if (set) PetscCall(airhip_push_option(ctx, names[i], val[0] ? val : "1"));
  }
  if (!ctx->is_inv) {
    /* the coarse-grid options the reference reaches through PCMG's own prefix (tests/Makefile:421-424, :147-149) */
    PetscCall(PetscOptionsString("-mg_coarse_ksp_max_it", "forwarded to libairhip", "", "", val, sizeof val, &set));
    if (set) PetscCall(airhip_push_option(ctx, "mg_coarse_ksp_max_it", val));
    PetscCall(PetscOptionsString("-mg_coarse_pc_type", "forwarded to libairhip", "", "", val, sizeof val, &set));
    if (set) PetscCall(airhip_push_option(ctx, "mg_coarse_pc_type", val));
  }
  PetscOptionsHeadEnd();
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetUp_AIR_c (src/PCAIR.c:182-201) / PCSetUp_PFLAREINV_c (src/PCPFLAREINV.c:689-783).  Rebuild policy: a frozen
 * preconditioner (pc->reusepreconditioner) stays as it is; otherwise any change of the operator rebuilds - the
 * library has no sparsity-reuse path, so SAME_NONZERO_PATTERN is treated like DIFFERENT_NONZERO_PATTERN */
#if defined(AIRHIP_WITH_MPI)
/* MPIAIJ: this rank's rows as CSR with GLOBAL, ascending column indices - the diagonal block's columns shifted by the
 * ownership start, the off-diagonal block's through its column map (what the reference reads with MatMPIAIJGetSeqAIJ,
 * src/PMISR_Module.F90:174-180) - handed to airhip_mpi_pcair_setup */
static PetscErrorCode PCSetUp_AIRHIP_MPI(PC pc)
{
  PC_AIRHIP         *ctx = (PC_AIRHIP *)pc->data;
  Mat                A   = pc->pmat, Ad, Ao;
  const PetscInt    *garray, *ia_d, *ja_d, *ia_o, *ja_o;
  const PetscScalar *va_d, *va_o;
  PetscInt           m, nloc, nd, no, cstart, cend, N;
  PetscBool          done_d, done_o, is_mpiaij;
  int64_t           *rp = NULL;
  int32_t           *cj = NULL;
  double            *vv = NULL;
  const char        *names[AIRHIP_MAX_OPTS], *values[AIRHIP_MAX_OPTS];
  airhip_csr         csr;
  int                rc;

  PetscFunctionBegin;
  PetscCheck(!ctx->is_inv, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "PCPFLAREINV on libairhip runs on one rank");
  PetscCall(PetscObjectBaseTypeCompare((PetscObject)A, MATMPIAIJ, &is_mpiaij));
  PetscCheck(is_mpiaij, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "the operator must be MATMPIAIJ (or a device subclass)");
  PetscCall(MatMPIAIJGetSeqAIJ(A, &Ad, &Ao, &garray));
  PetscCall(MatGetOwnershipRangeColumn(A, &cstart, &cend));
  PetscCall(MatGetSize(A, NULL, &N));
  PetscCall(MatGetLocalSize(A, &m, &nloc));
  PetscCall(MatGetRowIJ(Ad, 0, PETSC_FALSE, PETSC_FALSE, &nd, &ia_d, &ja_d, &done_d));
  PetscCall(MatGetRowIJ(Ao, 0, PETSC_FALSE, PETSC_FALSE, &no, &ia_o, &ja_o, &done_o));
  PetscCheck(done_d && done_o && nd == m && no == m, PETSC_COMM_SELF, PETSC_ERR_SUP, "operator blocks do not expose CSR arrays (MatGetRowIJ)");
  PetscCall(MatSeqAIJGetArrayRead(Ad, &va_d));
  PetscCall(MatSeqAIJGetArrayRead(Ao, &va_o));
  PetscCall(PetscMalloc1(m + 1, &rp));
  PetscCall(PetscMalloc1(ia_d[m] + ia_o[m] + 1, &cj));
  PetscCall(PetscMalloc1(ia_d[m] + ia_o[m] + 1, &vv));
  rp[0] = 0;
  for (PetscInt i = 0; i < m; i++) {
    /* merge: off-diagonal columns below cstart, the diagonal block, off-diagonal columns from cend on (garray ascends) */
    int64_t  p  = rp[i];
    PetscInt ko = ia_o[i];
    for (; ko < ia_o[i + 1] && garray[ja_o[ko]] < cstart; ko++) { cj[p] = (int32_t)garray[ja_o[ko]]; vv[p++] = (double)va_o[ko]; }
    for (PetscInt kd = ia_d[i]; kd < ia_d[i + 1]; kd++) { cj[p] = (int32_t)(ja_d[kd] + cstart); vv[p++] = (double)va_d[kd]; }
    for (; ko < ia_o[i + 1]; ko++) { cj[p] = (int32_t)garray[ja_o[ko]]; vv[p++] = (double)va_o[ko]; }
    rp[i + 1] = p;
  }
  PetscCall(MatSeqAIJRestoreArrayRead(Ao, &va_o));
  PetscCall(MatSeqAIJRestoreArrayRead(Ad, &va_d));
  PetscCall(MatRestoreRowIJ(Ao, 0, PETSC_FALSE, PETSC_FALSE, &no, &ia_o, &ja_o, &done_o));
  PetscCall(MatRestoreRowIJ(Ad, 0, PETSC_FALSE, PETSC_FALSE, &nd, &ia_d, &ja_d, &done_d));
  csr.m = m; csr.n = N; csr.rowptr = rp; csr.col = cj; csr.val = vv;
  for (int i = 0; i < ctx->nopts; i++) { names[i] = ctx->names[i]; values[i] = ctx->values[i]; }
  if (ctx->dist) (void)airhip_mpi_pcair_destroy(&ctx->dist);
  ctx->air = NULL;
  /* RCCL over xGMI when every rank has a GPU of its own (the reference's one-rank-per-GPU rule, docs/gpus.md) */
  rc = airhip_mpi_pcair_setup(PetscObjectComm((PetscObject)pc), &csr, ctx->nopts, names, values, NULL, AIRHIP_MPI_TRANSPORT_RCCL, 0, &ctx->dist, NULL);
  PetscCall(PetscFree(rp));
  PetscCall(PetscFree(cj));
  PetscCall(PetscFree(vv));
  PetscCheck(rc == AIRHIP_OK, PetscObjectComm((PetscObject)pc), PETSC_ERR_LIB, "libairhip_mpi (%d): %s", rc, airhip_mpi_last_error());
  rc = airhip_mpi_pcair_handle(ctx->dist, &ctx->air, NULL);
  PetscCheck(rc == AIRHIP_OK, PETSC_COMM_SELF, PETSC_ERR_LIB, "libairhip_mpi (%d): %s", rc, airhip_mpi_last_error());
  ctx->n = m;
  PetscFunctionReturn(PETSC_SUCCESS);
}
#endif

static PetscErrorCode PCSetUp_AIRHIP(PC pc)
{
  PC_AIRHIP         *ctx = (PC_AIRHIP *)pc->data;
  Mat                A   = pc->pmat; /* src/PCAIR.c:190 */
  PetscMPIInt        size;
  const PetscInt    *ia, *ja;
  PetscInt           n;
  PetscBool          done;
  const PetscScalar *va;
  int64_t           *rp  = NULL;
  int32_t           *cj  = NULL;
  airhip_csr         csr;
  int                rc;

  PetscFunctionBegin;
  if (pc->setupcalled && pc->reusepreconditioner) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCheck(sizeof(PetscScalar) == sizeof(double), PETSC_COMM_SELF, PETSC_ERR_SUP, "airhip computes in fp64 real arithmetic");
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)pc), &size));
  if (size > 1) {
#if defined(AIRHIP_WITH_MPI)
    PetscCall(PCSetUp_AIRHIP_MPI(pc));
    PetscFunctionReturn(PETSC_SUCCESS);
#else
    SETERRQ(PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "this build of the airhip glue sets up one rank; rebuild with -DAIRHIP_WITH_MPI (libairhip_mpi.so, INTEGRATION.md section 3)");
#endif
  }
  PetscCall(airhip_make_handle(ctx));
  PetscCall(MatGetRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done, PETSC_COMM_SELF, PETSC_ERR_SUP, "operator does not expose CSR arrays (MatGetRowIJ)");
  PetscCall(MatSeqAIJGetArrayRead(A, &va));
  /* the ABI takes int64 row pointers and int32 columns whatever PetscInt is */
  PetscCall(PetscMalloc1(n + 1, &rp));
  for (PetscInt i = 0; i <= n; i++) rp[i] = (int64_t)ia[i];
  if (sizeof(PetscInt) == sizeof(int32_t)) {
    csr.col = (const int32_t *)ja;
  } else {
    PetscCall(PetscMalloc1(ia[n], &cj));
    for (PetscInt k = 0; k < ia[n]; k++) cj[k] = (int32_t)ja[k];
    csr.col = cj;
  }
  csr.m = n; csr.n = n; csr.rowptr = rp; csr.val = (const double *)va;
  /* the library copies what it keeps; PETSc's arrays are released before the result is looked at, so that a failed
   * setup leaks nothing and leaves the Mat usable */
  rc = ctx->is_inv ? airhip_pcinv_setup(ctx->inv, &csr) : airhip_pcair_setup(ctx->air, &csr);
  PetscCall(PetscFree(rp));
  PetscCall(PetscFree(cj));
  PetscCall(MatSeqAIJRestoreArrayRead(A, &va));
  PetscCall(MatRestoreRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(rc == AIRHIP_OK, PETSC_COMM_SELF, PETSC_ERR_LIB, "airhip (%d): %s", rc, airhip_last_error());
  ctx->n = n;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCApply_AIR_c (src/PCAIR.c:150-166) / PCApply_PFLAREINV_c (src/PCPFLAREINV.c:618-626).
 * Device vectors (VECHIP, VECKOKKOS on HIP): their device pointers go straight to the library, ordered on PETSc's
 * current stream through the *_on_stream entry point - no host synchronisation per PCApply.  Host vectors are
 * staged through a device buffer. */
static PetscErrorCode PCApply_AIRHIP(PC pc, Vec x, Vec y)
{
  PC_AIRHIP         *ctx = (PC_AIRHIP *)pc->data;
  const PetscScalar *px;
  PetscScalar       *py;
  PetscMemType       mx, my;
  void              *stream = NULL;
  int                have_stream = 0;

  PetscFunctionBegin;
#if PETSC_VERSION_GE(3, 20, 0) && !defined(AIRHIP_PETSC_NO_STREAM_HANDLE)
  {
    /* the stream PETSc's current device context launches on: the library orders itself behind it and makes it wait
     * for the result (no host synchronisation per PCApply) */
    PetscDeviceContext dctx;
    void              *handle;
    PetscCall(PetscDeviceContextGetCurrentContext(&dctx));
    PetscCall(PetscDeviceContextGetStreamHandle(dctx, &handle));
    stream      = *(void **)handle;
    have_stream = 1;
  }
#endif
  PetscCall(VecGetArrayReadAndMemType(x, &px, &mx));
  PetscCall(VecGetArrayWriteAndMemType(y, &py, &my));
  if (PetscMemTypeDevice(mx) && PetscMemTypeDevice(my)) {
    /* no stream handle from this PETSc: whatever stream produced x is unknown, so wait for the device before the apply
     * and for the library after it - correct under every stream policy, one host synchronisation per PCApply */
    if (!have_stream) PetscCheck(hipDeviceSynchronize() == hipSuccess, PETSC_COMM_SELF, PETSC_ERR_LIB, "hipDeviceSynchronize");
    if (ctx->is_inv) AIRHIP_CALL(airhip_pcinv_apply_on_stream(ctx->inv, (const double *)px, (double *)py, stream));
    else AIRHIP_CALL(airhip_pcair_apply_on_stream(ctx->air, (const double *)px, (double *)py, stream));
    if (!have_stream) {
      if (ctx->is_inv) AIRHIP_CALL(airhip_pcinv_sync(ctx->inv));
      else AIRHIP_CALL(airhip_pcair_sync(ctx->air));
    }
  } else {
    const size_t bytes = (size_t)ctx->n * sizeof(double);
    if (!ctx->d_x) {
      PetscCheck(hipMalloc((void **)&ctx->d_x, bytes) == hipSuccess && hipMalloc((void **)&ctx->d_y, bytes) == hipSuccess,
                 PETSC_COMM_SELF, PETSC_ERR_MEM, "hipMalloc of the staging vectors failed");
    }
    PetscCheck(hipMemcpy(ctx->d_x, px, bytes, hipMemcpyDefault) == hipSuccess, PETSC_COMM_SELF, PETSC_ERR_LIB, "hipMemcpy");
    if (ctx->is_inv) {
      AIRHIP_CALL(airhip_pcinv_apply(ctx->inv, ctx->d_x, ctx->d_y));
      AIRHIP_CALL(airhip_pcinv_sync(ctx->inv));
    } else {
      AIRHIP_CALL(airhip_pcair_apply(ctx->air, ctx->d_x, ctx->d_y));
      AIRHIP_CALL(airhip_pcair_sync(ctx->air));
    }
    PetscCheck(hipMemcpy(py, ctx->d_y, bytes, hipMemcpyDefault) == hipSuccess, PETSC_COMM_SELF, PETSC_ERR_LIB, "hipMemcpy");
  }
  PetscCall(VecRestoreArrayWriteAndMemType(y, &py));
  PetscCall(VecRestoreArrayReadAndMemType(x, &px));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCReset_AIR_c / PCReset_PFLAREINV_c: free the hierarchy, keep the options */
static PetscErrorCode PCReset_AIRHIP(PC pc)
{
  PC_AIRHIP *ctx = (PC_AIRHIP *)pc->data;

  PetscFunctionBegin;
#if defined(AIRHIP_WITH_MPI)
  if (ctx->dist) {
    (void)airhip_mpi_pcair_destroy(&ctx->dist); /* owns the rank-local PCAIR `air` points to */
    ctx->air = NULL;
  }
#endif
  if (ctx->air) AIRHIP_CALL(airhip_pcair_destroy(&ctx->air));
  if (ctx->inv) AIRHIP_CALL(airhip_pcinv_destroy(&ctx->inv));
  if (ctx->d_x) (void)hipFree(ctx->d_x);
  if (ctx->d_y) (void)hipFree(ctx->d_y);
  ctx->d_x = ctx->d_y = NULL;
  ctx->n = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_AIRHIP(PC pc)
{
  PetscFunctionBegin;
  PetscCall(PCReset_AIRHIP(pc));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCView_AIR_c (src/PCAIR.c) / -pc_air_print_stats_timings output (src/AIR_MG_Stats.F90:386-416) */
static PetscErrorCode PCView_AIRHIP(PC pc, PetscViewer viewer)
{
  PC_AIRHIP *ctx = (PC_AIRHIP *)pc->data;
  PetscBool  ascii;

  PetscFunctionBegin;
  PetscCall(PetscObjectTypeCompare((PetscObject)viewer, PETSCVIEWERASCII, &ascii));
  if (!ascii) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscViewerASCIIPrintf(viewer, "  %s on libairhip (MI355X / gfx950), version %d\n", ctx->is_inv ? "PCPFLAREINV" : "PCAIR", airhip_version()));
  for (int i = 0; i < ctx->nopts; i++) PetscCall(PetscViewerASCIIPrintf(viewer, "    %s = %s\n", ctx->names[i], ctx->values[i]));
  if (!ctx->is_inv && ctx->air && ctx->n > 0) {
    int    levels = 0;
    double grid, op, cycle, storage, bytes;
    AIRHIP_CALL(airhip_pcair_num_levels(ctx->air, &levels));
    AIRHIP_CALL(airhip_pcair_complexities(ctx->air, &grid, &op, &cycle, &storage, &bytes));
    PetscCall(PetscViewerASCIIPrintf(viewer, "    levels %d, grid complexity %g, operator complexity %g, cycle complexity %g, storage complexity %g\n",
                                     levels, grid, op, cycle, storage));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* programmatic twin of the options database: PCAIRSetOption(pc, "strong_threshold", "0.25") - the typed
 * PCAIRSet* / PCAIRGet* families of include/pflare.h:121-243 map onto it one line each */
static PetscErrorCode PCAIRSetOption_AIRHIP(PC pc, const char *name, const char *value)
{
  PetscFunctionBegin;
  PetscCall(airhip_push_option((PC_AIRHIP *)pc->data, name, value));
  PetscFunctionReturn(PETSC_SUCCESS);
}


/* ---- typed entry points of include/pflare.h:124-244 (the subset whose options this library implements).  Setters store
 * the option like the options database does (effective at the next PCSetUp, as in the reference); the getters of
 * results read the hierarchy. */
#define AIRHIP_SETTER(fn, petsc_type, name, fmt, cast)                                \
  PETSC_EXTERN PetscErrorCode fn(PC pc, petsc_type v)                                 \
  {                                                                                   \
    char buf[64];                                                                     \
    PetscFunctionBegin;                                                               \
    PetscCall(PetscSNPrintf(buf, sizeof buf, fmt, (cast)v));                          \
    PetscCall(airhip_push_option((PC_AIRHIP *)pc->data, name, buf));                  \
    PetscFunctionReturn(PETSC_SUCCESS);                                               \
  }
static const char *const airhip_inv_names[] = {"power", "arnoldi", "newton", "newton_no_extra", "neumann", "sai", "isai", "wjacobi", "jacobi"}; /* include/pflare.h:36-46 */
static const char *const airhip_z_names[]   = {"product", "lair", "lair_sai"};                                                            /* include/pflare.h:66-70 */
#define AIRHIP_ENUM_SETTER(fn, name, table, count)                                                       \
  PETSC_EXTERN PetscErrorCode fn(PC pc, int v)                                                           \
  {                                                                                                      \
    PetscFunctionBegin;                                                                                  \
    PetscCheck(v >= 0 && v < (count), PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "enum value %d", v);    \
    PetscCall(airhip_push_option((PC_AIRHIP *)pc->data, name, table[v]));                                \
    PetscFunctionReturn(PETSC_SUCCESS);                                                                  \
  }
AIRHIP_SETTER(PCAIRSetPrintStatsTimings, PetscBool, "print_stats_timings", "%d", int)
AIRHIP_SETTER(PCAIRSetMaxLevels, PetscInt, "max_levels", "%d", int)
AIRHIP_SETTER(PCAIRSetCoarseEqLimit, PetscInt, "coarse_eq_limit", "%d", int)
AIRHIP_SETTER(PCAIRSetAutoTruncateStartLevel, PetscInt, "auto_truncate_start_level", "%d", int)
AIRHIP_SETTER(PCAIRSetAutoTruncateTol, PetscReal, "auto_truncate_tol", "%.17g", double)
AIRHIP_SETTER(PCAIRSetProcessorAgglom, PetscBool, "processor_agglom", "%d", int)
AIRHIP_SETTER(PCAIRSetStrongThreshold, PetscReal, "strong_threshold", "%.17g", double)
AIRHIP_SETTER(PCAIRSetDDCIts, PetscInt, "ddc_its", "%d", int)
AIRHIP_SETTER(PCAIRSetDDCFraction, PetscReal, "ddc_fraction", "%.17g", double)
AIRHIP_SETTER(PCAIRSetCFSplittingType, int, "cf_splitting_type", "%d", int)
AIRHIP_SETTER(PCAIRSetMaxLubySteps, PetscInt, "max_luby_steps", "%d", int)
AIRHIP_SETTER(PCAIRSetDiagScalePolys, PetscBool, "diag_scale_polys", "%d", int)
AIRHIP_SETTER(PCAIRSetMatrixFreePolys, PetscBool, "matrix_free_polys", "%d", int)
AIRHIP_SETTER(PCAIRSetOnePointClassicalProlong, PetscBool, "one_point_classical_prolong", "%d", int)
AIRHIP_SETTER(PCAIRSetFullSmoothingUpAndDown, PetscBool, "full_smoothing_up_and_down", "%d", int)
AIRHIP_SETTER(PCAIRSetSymmetric, PetscBool, "symmetric", "%d", int)
AIRHIP_SETTER(PCAIRSetImproveWIts, PetscInt, "improve_w_its", "%d", int)
AIRHIP_SETTER(PCAIRSetImproveZIts, PetscInt, "improve_z_its", "%d", int)
AIRHIP_SETTER(PCAIRSetStrongRThreshold, PetscReal, "strong_r_threshold", "%.17g", double)
AIRHIP_SETTER(PCAIRSetLairDistance, PetscInt, "lair_distance", "%d", int)
AIRHIP_SETTER(PCAIRSetPolyOrder, PetscInt, "poly_order", "%d", int)
AIRHIP_SETTER(PCAIRSetInverseSparsityOrder, PetscInt, "inverse_sparsity_order", "%d", int)
AIRHIP_SETTER(PCAIRSetCPolyOrder, PetscInt, "c_poly_order", "%d", int)
AIRHIP_SETTER(PCAIRSetCInverseSparsityOrder, PetscInt, "c_inverse_sparsity_order", "%d", int)
AIRHIP_SETTER(PCAIRSetCoarsestPolyOrder, PetscInt, "coarsest_poly_order", "%d", int)
AIRHIP_SETTER(PCAIRSetCoarsestInverseSparsityOrder, PetscInt, "coarsest_inverse_sparsity_order", "%d", int)
AIRHIP_SETTER(PCAIRSetCoarsestMatrixFreePolys, PetscBool, "coarsest_matrix_free_polys", "%d", int)
AIRHIP_SETTER(PCAIRSetCoarsestDiagScalePolys, PetscBool, "coarsest_diag_scale_polys", "%d", int)
AIRHIP_SETTER(PCAIRSetRDrop, PetscReal, "r_drop", "%.17g", double)
AIRHIP_SETTER(PCAIRSetADrop, PetscReal, "a_drop", "%.17g", double)
AIRHIP_SETTER(PCAIRSetALump, PetscBool, "a_lump", "%d", int)
AIRHIP_ENUM_SETTER(PCAIRSetInverseType, "inverse_type", airhip_inv_names, 9)
AIRHIP_ENUM_SETTER(PCAIRSetCInverseType, "c_inverse_type", airhip_inv_names, 9)
AIRHIP_ENUM_SETTER(PCAIRSetCoarsestInverseType, "coarsest_inverse_type", airhip_inv_names, 9)
AIRHIP_ENUM_SETTER(PCAIRSetZType, "z_type", airhip_z_names, 3)
AIRHIP_ENUM_SETTER(PCPFLAREINVSetType, "type", airhip_inv_names, 9)
AIRHIP_SETTER(PCPFLAREINVSetPolyOrder, PetscInt, "poly_order", "%d", int)
AIRHIP_SETTER(PCPFLAREINVSetSparsityOrder, PetscInt, "sparsity_order", "%d", int)
AIRHIP_SETTER(PCPFLAREINVSetMatrixFree, PetscBool, "matrix_free", "%d", int)

PETSC_EXTERN PetscErrorCode PCAIRSetSmoothType(PC pc, const char *v)
{
  PetscFunctionBegin;
  PetscCall(airhip_push_option((PC_AIRHIP *)pc->data, "smooth_type", v));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode PCAIRGetNumLevels(PC pc, PetscInt *levels)
{
  PC_AIRHIP *ctx = (PC_AIRHIP *)pc->data;
  int        l   = 0;

  PetscFunctionBegin;
  PetscCheck(ctx->air, PETSC_COMM_SELF, PETSC_ERR_ORDER, "PCSetUp first");
  AIRHIP_CALL(airhip_pcair_num_levels(ctx->air, &l));
  *levels = l;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCAIRGetGridComplexity / OperatorComplexity / CycleComplexity / StorageComplexity (src/AIR_MG_Stats.F90:79-382) */
static PetscErrorCode airhip_complexity(PC pc, int which, PetscReal *out)
{
  PC_AIRHIP *ctx = (PC_AIRHIP *)pc->data;
  double     c[5];

  PetscFunctionBegin;
  PetscCheck(ctx->air, PETSC_COMM_SELF, PETSC_ERR_ORDER, "PCSetUp first");
  AIRHIP_CALL(airhip_pcair_complexities(ctx->air, &c[0], &c[1], &c[2], &c[3], &c[4]));
  *out = (PetscReal)c[which];
  PetscFunctionReturn(PETSC_SUCCESS);
}
PETSC_EXTERN PetscErrorCode PCAIRGetGridComplexity(PC pc, PetscReal *v) { return airhip_complexity(pc, 0, v); }
PETSC_EXTERN PetscErrorCode PCAIRGetOperatorComplexity(PC pc, PetscReal *v) { return airhip_complexity(pc, 1, v); }
PETSC_EXTERN PetscErrorCode PCAIRGetCycleComplexity(PC pc, PetscReal *v) { return airhip_complexity(pc, 2, v); }
PETSC_EXTERN PetscErrorCode PCAIRGetStorageComplexity(PC pc, PetscReal *v) { return airhip_complexity(pc, 3, v); }

static PetscErrorCode PCCreate_AIRHIP(PC pc, int is_inv)
{
  PC_AIRHIP *ctx;

  PetscFunctionBegin;
  PetscCall(PetscNew(&ctx));
  ctx->is_inv             = is_inv;
  pc->data                = (void *)ctx;
  pc->ops->apply          = PCApply_AIRHIP;
  pc->ops->setup          = PCSetUp_AIRHIP;
  pc->ops->destroy        = PCDestroy_AIRHIP;
  pc->ops->view           = PCView_AIRHIP;
  pc->ops->reset          = PCReset_AIRHIP;
  pc->ops->setfromoptions = PCSetFromOptions_AIRHIP;
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, is_inv ? "PCPFLAREINVSetOption_C" : "PCAIRSetOption_C", PCAIRSetOption_AIRHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode PCCreate_AIR(PC pc)
{
  PetscFunctionBegin;
  PetscCall(PCCreate_AIRHIP(pc, 0));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode PCCreate_PFLAREINV(PC pc)
{
  PetscFunctionBegin;
  PetscCall(PCCreate_AIRHIP(pc, 1));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN void PCRegister_AIR(void)
{
  PetscCallVoid(PCRegister("air", PCCreate_AIR));
}

PETSC_EXTERN void PCRegister_PFLAREINV(void)
{
  PetscCallVoid(PCRegister("pflareinv", PCCreate_PFLAREINV));
}

/* called by PETSc when the shared library is loaded (-dll_append / PetscDLLibraryAppend) */
PETSC_EXTERN PetscErrorCode PetscDLLibraryRegister_petscpflare(void)
{
  PetscFunctionBegin;
  PCRegister_AIR();
  PCRegister_PFLAREINV();
  PetscFunctionReturn(PETSC_SUCCESS);
}
