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
 * Scope: one MPI rank (MATSEQAIJ and its device subclasses).  For MPIAIJ the per-rank hierarchy pieces go through
 * airhip_partition_* / airhip_pcair_load_distributed; this file refuses communicators
 * of size > 1 instead of silently solving a different problem.
 */
#include <petsc/private/pcimpl.h>
#include <petscmat.h>
#include <petscvec.h>
#include <petscdevice.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "airhip.h"

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
static PetscErrorCode PCSetFromOptions_AIRHIP(PC pc, PetscOptionItems PetscOptionsObject)
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
static PetscErrorCode PCSetUp_AIRHIP(PC pc)
{
  PC_AIRHIP         *ctx = (PC_AIRHIP *)pc->data;
  Mat                A   = pc->pmat; /* src/PCAIR.c:190 */
  PetscMPIInt        size;
  const PetscInt    *ia, *ja;
  PetscInt           n;
  PetscBool          done;
  const PetscScalar *va;
  int64_t           *rp;
  int32_t           *cj  = NULL;
  airhip_csr         csr;

  PetscFunctionBegin;
  if (pc->setupcalled && pc->reusepreconditioner) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)pc), &size));
  PetscCheck(size == 1, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP,
             "the airhip glue sets up one rank; distribute with airhip_partition_* (INTEGRATION.md section 3)");
  PetscCheck(sizeof(PetscScalar) == sizeof(double), PETSC_COMM_SELF, PETSC_ERR_SUP, "airhip computes in fp64 real arithmetic");
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
  if (ctx->is_inv) AIRHIP_CALL(airhip_pcinv_setup(ctx->inv, &csr));
  else AIRHIP_CALL(airhip_pcair_setup(ctx->air, &csr)); /* copies what it keeps; PETSc's arrays are released below */
  ctx->n = n;
  PetscCall(PetscFree(rp));
  PetscCall(PetscFree(cj));
  PetscCall(MatSeqAIJRestoreArrayRead(A, &va));
  PetscCall(MatRestoreRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
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
  void              *stream = NULL; /* NULL = the legacy default stream (PETSC_STREAM_DEFAULT) */

  PetscFunctionBegin;
#if defined(AIRHIP_PETSC_STREAM_HANDLE)
  {
    /* PETSc builds that expose the stream of the current device context (PetscDeviceContextGetStreamHandle) */
    PetscDeviceContext dctx;
    void              *handle;
    PetscCall(PetscDeviceContextGetCurrentContext(&dctx));
    PetscCall(PetscDeviceContextGetStreamHandle(dctx, &handle));
    stream = *(void **)handle;
  }
#endif
  PetscCall(VecGetArrayReadAndMemType(x, &px, &mx));
  PetscCall(VecGetArrayWriteAndMemType(y, &py, &my));
  if (PetscMemTypeDevice(mx) && PetscMemTypeDevice(my)) {
    if (ctx->is_inv) AIRHIP_CALL(airhip_pcinv_apply_on_stream(ctx->inv, (const double *)px, (double *)py, stream));
    else AIRHIP_CALL(airhip_pcair_apply_on_stream(ctx->air, (const double *)px, (double *)py, stream));
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
