/* MINIMAL MOCK of the subset of the PETSc C API that pflare_amd/csrc/petsc_glue.c uses - TEST INFRASTRUCTURE.
 * PETSc is not available in the image this repository is developed in, so the glue cannot be built for real; this mock lets
 * tests/test_petsc_glue.py run the C compiler over it (syntax, types, argument counts, macro expansions, internal
 * consistency with include/airhip.h).  It was written from the PETSc manual pages' prototypes and proves nothing about
 * compatibility with a real PETSc: declarations only, nothing here is ever linked or executed. */
#ifndef PETSCSYS_MOCK_H
#define PETSCSYS_MOCK_H
#include <mpi.h>
#include <stddef.h>
#include <stdio.h>

#define PETSC_VERSION_MAJOR 3
#ifndef PETSC_VERSION_MINOR
#define PETSC_VERSION_MINOR 25
#endif
#define PETSC_VERSION_GE(a, b, c) ((PETSC_VERSION_MAJOR > (a)) || (PETSC_VERSION_MAJOR == (a) && PETSC_VERSION_MINOR >= (b)))
#define PETSC_EXTERN extern

typedef int PetscErrorCode;
typedef int PetscInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscMPIInt;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { PETSC_MEMTYPE_HOST = 0, PETSC_MEMTYPE_DEVICE = 1, PETSC_MEMTYPE_HIP = 3 } PetscMemType;
#define PetscMemTypeDevice(m) (((m) & 0x1) == PETSC_MEMTYPE_DEVICE)
#define PETSC_SUCCESS 0
#define PETSC_ERR_LIB 76
#define PETSC_ERR_SUP 56
#define PETSC_ERR_MEM 55
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_COMM_SELF MPI_COMM_SELF

typedef struct _p_PetscObject *PetscObject;
typedef struct _p_Mat *Mat;
typedef struct _p_Vec *Vec;
typedef struct _p_PC *PC;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscDeviceContext *PetscDeviceContext;
typedef struct _p_PetscOptionItems *PetscOptionItems;
typedef const char *MatType;
typedef const char *PetscViewerType;
#define MATMPIAIJ "mpiaij"
#define PETSCVIEWERASCII "ascii"

PetscErrorCode PetscErrorMock(MPI_Comm, int, const char *, ...);
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(x) return (x)
#define PetscCall(...) do { PetscErrorCode ierr_mock_ = (__VA_ARGS__); if (ierr_mock_) return ierr_mock_; } while (0)
#define PetscCallMPI(...) do { int ierr_mock_ = (__VA_ARGS__); if (ierr_mock_) return 1; } while (0)
#define PetscCallVoid(...) do { PetscErrorCode ierr_mock_ = (__VA_ARGS__); if (ierr_mock_) return; } while (0)
#define PetscCheck(cond, comm, code, ...) do { if (!(cond)) return PetscErrorMock(comm, code, __VA_ARGS__); } while (0)
#define SETERRQ(comm, code, ...) return PetscErrorMock(comm, code, __VA_ARGS__)

PetscErrorCode PetscMallocMock(size_t, void *);
#define PetscMalloc1(n, p) PetscMallocMock((size_t)(n) * sizeof(**(p)), (void *)(p))
#define PetscNew(p) PetscMallocMock(sizeof(**(p)), (void *)(p))
PetscErrorCode PetscFreeMock(void *);
#define PetscFree(p) PetscFreeMock((void *)(p))
PetscErrorCode PetscSNPrintf(char *, size_t, const char *, ...);
PetscErrorCode PetscObjectGetComm(PetscObject, MPI_Comm *);
MPI_Comm PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char *, PetscBool *);
PetscErrorCode PetscObjectBaseTypeCompare(PetscObject, const char *, PetscBool *);
PetscErrorCode PetscObjectComposeFunction_Private(PetscObject, const char *, void (*)(void));
#define PetscObjectComposeFunction(o, n, f) PetscObjectComposeFunction_Private((PetscObject)(o), (n), (void (*)(void))(f))

/* options */
#if PETSC_VERSION_GE(3, 23, 0)
typedef PetscOptionItems PetscOptionItemsArg;      /* passed by value from 3.23 on */
#else
typedef PetscOptionItems *PetscOptionItemsArg;     /* by pointer before */
#endif
PetscErrorCode PetscOptionsHeadBeginMock(PetscOptionItemsArg, const char *);
#define PetscOptionsHeadBegin(obj, head) do { PetscCall(PetscOptionsHeadBeginMock(obj, head)); } while (0)
#define PetscOptionsHeadEnd() do { } while (0)
PetscErrorCode PetscOptionsStringMock(PetscOptionItemsArg, const char *, const char *, const char *, const char *, char *, size_t, PetscBool *);
#define PetscOptionsString(opt, text, man, cur, val, len, set) PetscOptionsStringMock(PetscOptionsObject, opt, text, man, cur, val, len, set)

/* Mat */
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt **, const PetscInt **, PetscBool *);
PetscErrorCode MatRestoreRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt **, const PetscInt **, PetscBool *);
PetscErrorCode MatSeqAIJGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqAIJRestoreArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatMPIAIJGetSeqAIJ(Mat, Mat *, Mat *, const PetscInt **);
PetscErrorCode MatGetOwnershipRangeColumn(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetLocalSize(Mat, PetscInt *, PetscInt *);
/* Vec */
PetscErrorCode VecGetArrayReadAndMemType(Vec, const PetscScalar **, PetscMemType *);
PetscErrorCode VecGetArrayWriteAndMemType(Vec, PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayReadAndMemType(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArrayWriteAndMemType(Vec, PetscScalar **);
/* device context */
PetscErrorCode PetscDeviceContextGetCurrentContext(PetscDeviceContext *);
PetscErrorCode PetscDeviceContextGetStreamHandle(PetscDeviceContext, void **);
/* viewer */
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer, const char *, ...);
/* PC */
struct _PCOps {
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*view)(PC, PetscViewer);
  PetscErrorCode (*reset)(PC);
#if PETSC_VERSION_GE(3, 23, 0)
  PetscErrorCode (*setfromoptions)(PC, PetscOptionItems);
#else
  PetscErrorCode (*setfromoptions)(PC, PetscOptionItems *);
#endif
};
struct _p_PC {
  struct _PCOps ops[1];
  void *data;
  Mat mat, pmat;
  PetscBool setupcalled, reusepreconditioner;
  int flag;
};
PetscErrorCode PCRegister(const char *, PetscErrorCode (*)(PC));
#endif
