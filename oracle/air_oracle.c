/* air_oracle.c - plain C restatement of the PCAIR *solve phase* (CPU).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker: only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may build, load or call it.  Nothing in pflare_amd/ links
 * against it and the GPU product path never routes through it.
 *
 * It follows the reference's solve-phase code operation by operation, in the reference's
 * level-local ("natural") numbering with explicit F/C gathers and scatters:
 *   mg_coarse_shell_apply        src/FC_Smooth.F90:29-49
 *   VecISCopy gather / scatter   src/FC_Smooth.F90:161-417, src/VecISCopyLocalk.kokkos.cxx:137-195
 *   mg_FC_point_richardson       src/FC_Smooth.F90:421-495
 *   f_smooths / c_smooths        src/FC_Smooth.F90:499-640
 *   petsc_horner                 src/Gmres_Poly.F90:1418-1484
 *   D^-1 A / right-scaled shells src/Gmres_Poly.F90:1341-1414
 *   petsc_newton                 src/Gmres_Poly_Newton.F90:763-875 (+ right-scaled :716-759)
 *   Neumann shell                src/Neumann_Poly.F90:19-55
 *   Kaskade cycle                PETSc PCMGKCycle_Private as wired by src/AIR_MG_Setup.F90:967-1156
 *   KSPRICHARDSON / KSPGMRES     PETSc semantics (CGS, restart, KSPConvergedDefault)
 *
 * PARITY STATUS: the reference cannot be built here (PETSc + Fortran), so this restatement is
 * pinned through tests/test_oracle_reference_bounds.py (the reference's own iteration-count
 * bounds and exact-solver properties) and cross-checked against oracle/air_ref.py (scipy).
 *
 * Struct layouts of oc_csr / oc_inv equal airhip_csr / airhip_invop so tests can pass the
 * very same host arrays the product exported (zero copy).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { int64_t m, n; const int64_t* rp; const int32_t* ci; const double* v; } oc_csr;
typedef struct {
  int kind;            /* 0 none, 1 assembled, 2 diag, 3 horner, 4 newton */
  oc_csr mat;
  const double* diag;
  int ncoef; const double* coef; const double* coef_im;
  int diag_scale, neumann;
} oc_inv;

typedef struct {
  int64_t n, n_f, n_c;
  const int32_t *F, *C;
  int has_aff; oc_csr A_ff; const double* aff_diag;
  oc_csr A_fc, Z;
  int one_point; const int32_t* wcol; oc_csr W;
  int has_c; oc_csr A_cf, A_cc; oc_inv inv_ff, inv_cc;
  int smooth_order[10];
  /* work */
  double *b, *x, *xf, *xc, *t2, *t3, *t4, *c2, *c3, *c4;
  double *m1, *m2, *m3, *mrhs, *dff, *dcc;
} oc_level;

typedef struct {
  int nlev;               /* number of levels including the coarsest */
  oc_level* L;            /* nlev-1 entries */
  oc_csr A_c; oc_inv inv_c; int rich_its;
  double *bc, *xc, *rc, *m1, *m2, *m3, *mrhs, *dc;
  int nthreads;
} oc_hier;

static const double TOL_ZERO = (double)1e-12f; /* src/Pflare_Parameters.F90:206 */

static void spmv(const oc_csr* A, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A->m; i++) {
    double s = 0.0;
    for (int64_t k = A->rp[i]; k < A->rp[i + 1]; k++) s += A->v[k] * x[A->ci[k]];
    y[i] = s;
  }
}
static void vcopy(int64_t n, const double* a, double* b) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) b[i] = a[i];
}
/* y = a*x + b*y  (VecAXPBY) */
static void axpby(int64_t n, double a, const double* x, double b, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = a * x[i] + b * y[i];
}
/* y += a*x (VecAXPY) */
static void axpy(int64_t n, double a, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] += a * x[i];
}
/* y = x + a*y (VecAYPX) */
static void aypx(int64_t n, double a, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = x[i] + a * y[i];
}
static void gather(int64_t n, const int32_t* idx, const double* full, double* red) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) red[i] = full[idx[i]];
}
static void scatter(int64_t n, const int32_t* idx, const double* red, double* full) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) full[idx[i]] = red[i];
}
static double dot(int64_t n, const double* a, const double* b) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

/* operator handle for the matrix-free polynomials: CSR or diagonal */
typedef struct { const oc_csr* A; const double* d; int64_t n; const double* diag; int dscale, neumann; } opv;

static void op_apply(const opv* o, const double* x, double* y) {
  if (o->A) spmv(o->A, x, y);
  else {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < o->n; i++) y[i] = o->d[i] * x[i];
  }
  if (o->dscale) {           /* petsc_matvec_da_poly_mf: y = D^-1 (A x) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < o->n; i++) y[i] = y[i] / o->diag[i];
    if (o->neumann)          /* petsc_matvec_ida_neumann_poly_mf: y = x - D^-1 A x */
      axpby(o->n, 1.0, x, -1.0, y);
  }
}

/* petsc_horner */
static void horner(const opv* o, const double* c, int nc, double* tmp, const double* x, double* y) {
  const int64_t n = o->n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = c[nc - 1] * x[i];
  for (int k = nc - 2; k >= 0; k--) {
    if (c[k] == 0.0) continue;
    vcopy(n, y, tmp);
    op_apply(o, tmp, y);
    axpby(n, c[k], x, 1.0, y);
  }
}

/* petsc_newton */
static void newton(const opv* o, const double* re, const double* im, int nr, double* t, double* t2, double* t3,
                   const double* x, double* y) {
  const int64_t n = o->n;
  vcopy(n, x, t);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = 0.0;
  int i = 0;
  while (i <= nr - 2) {
    if (im[i] == 0.0) {
      if (fabs(re[i]) < TOL_ZERO) { i++; continue; }
      axpy(n, 1.0 / re[i], t, y);
      op_apply(o, t, t2);
      axpy(n, -1.0 / re[i], t2, t);
      i++;
    } else {
      double sq = re[i] * re[i] + im[i] * im[i];
      if (sq < TOL_ZERO) { i += 2; continue; }
      op_apply(o, t, t2);
      axpby(n, 2 * re[i], t, -1.0, t2);
      axpy(n, 1.0 / sq, t2, y);
      if (i <= nr - 3) { op_apply(o, t2, t3); axpy(n, -1.0 / sq, t3, t); }
      i += 2;
    }
  }
  if (im[nr - 1] == 0.0 && fabs(re[nr - 1]) > TOL_ZERO) axpby(n, 1.0 / re[nr - 1], t, 1.0, y);
}

/* y = M x */
static void apply_inv(const oc_inv* I, const oc_csr* A, const double* adiag_mat, int64_t n, const double* opdiag,
                      double* m1, double* m2, double* m3, double* mrhs, const double* x, double* y) {
  if (I->kind == 1) { spmv(&I->mat, x, y); return; }
  if (I->kind == 2) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = I->diag[i] * x[i];
    return;
  }
  opv o; o.A = A; o.d = adiag_mat; o.n = n; o.diag = opdiag; o.dscale = I->diag_scale; o.neumann = I->neumann;
  const double* r = x;
  if (I->diag_scale) {   /* MF_VEC_RHS = D^-1 x */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) mrhs[i] = x[i] / opdiag[i];
    r = mrhs;
  }
  if (I->kind == 3) horner(&o, I->coef, I->ncoef, m1, r, y);
  else newton(&o, I->coef, I->coef_im, I->ncoef, m1, m2, m3, r, y);
}

static double* dalloc(int64_t n) { return (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); }

static void csr_diag(const oc_csr* A, double* d) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A->m; i++) {
    double x = 0.0;
    for (int64_t k = A->rp[i]; k < A->rp[i + 1]; k++) if (A->ci[k] == i) { x = A->v[k]; break; }
    d[i] = x;
  }
}

oc_hier* oc_create(int nlev) {
  oc_hier* h = (oc_hier*)calloc(1, sizeof(oc_hier));
  h->nlev = nlev;
  h->L = (oc_level*)calloc(nlev > 1 ? nlev - 1 : 1, sizeof(oc_level));
  return h;
}

int oc_set_level(oc_hier* h, int l, int64_t n, int64_t n_f, int64_t n_c, const int32_t* F, const int32_t* C,
                 const oc_csr* A_ff, const double* aff_diag, const oc_csr* A_fc, const oc_inv* inv_ff, const oc_csr* Z,
                 const int32_t* wcol, const oc_csr* W, const oc_csr* A_cf, const oc_csr* A_cc, const oc_inv* inv_cc,
                 const int* smooth_order) {
  if (!h || l < 0 || l >= h->nlev - 1) return -1;
  oc_level* L = &h->L[l];
  L->n = n; L->n_f = n_f; L->n_c = n_c; L->F = F; L->C = C;
  L->has_aff = A_ff != NULL; if (A_ff) L->A_ff = *A_ff; L->aff_diag = aff_diag;
  L->A_fc = *A_fc; L->Z = *Z; L->inv_ff = *inv_ff;
  L->one_point = wcol != NULL; L->wcol = wcol; if (W) L->W = *W;
  L->has_c = 0;
  for (int i = 0; i < 10; i++) { L->smooth_order[i] = smooth_order[i]; if (smooth_order[i] < 0) L->has_c = 1; }
  if (L->has_c) { if (!A_cf || !A_cc || !inv_cc) return -2; L->A_cf = *A_cf; L->A_cc = *A_cc; L->inv_cc = *inv_cc; }
  L->b = dalloc(n); L->x = dalloc(n);
  L->xf = dalloc(n_f); L->t2 = dalloc(n_f); L->t3 = dalloc(n_f); L->t4 = dalloc(n_f);
  L->xc = dalloc(n_c); L->c2 = dalloc(n_c); L->c3 = dalloc(n_c); L->c4 = dalloc(n_c);
  int64_t nm = n_f > n_c ? n_f : n_c;
  L->m1 = dalloc(nm); L->m2 = dalloc(nm); L->m3 = dalloc(nm); L->mrhs = dalloc(nm);
  L->dff = dalloc(n_f); L->dcc = dalloc(n_c);
  if (L->has_aff) csr_diag(&L->A_ff, L->dff); else memcpy(L->dff, aff_diag, sizeof(double) * (size_t)n_f);
  if (L->has_c) csr_diag(&L->A_cc, L->dcc);
  return 0;
}

int oc_set_coarse(oc_hier* h, const oc_csr* A, const oc_inv* inv, int rich_its) {
  h->A_c = *A; h->inv_c = *inv; h->rich_its = rich_its;
  int64_t n = A->m;
  h->bc = dalloc(n); h->xc = dalloc(n); h->rc = dalloc(n);
  h->m1 = dalloc(n); h->m2 = dalloc(n); h->m3 = dalloc(n); h->mrhs = dalloc(n); h->dc = dalloc(n);
  csr_diag(A, h->dc);
  return 0;
}

void oc_destroy(oc_hier* h) {
  if (!h) return;
  for (int l = 0; l < h->nlev - 1; l++) {
    oc_level* L = &h->L[l];
    free(L->b); free(L->x); free(L->xf); free(L->t2); free(L->t3); free(L->t4); free(L->xc); free(L->c2); free(L->c3);
    free(L->c4); free(L->m1); free(L->m2); free(L->m3); free(L->mrhs); free(L->dff); free(L->dcc);
  }
  free(h->L); free(h->bc); free(h->xc); free(h->rc); free(h->m1); free(h->m2); free(h->m3); free(h->mrhs); free(h->dc);
  free(h);
}

int oc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}

static void aff_mult(const oc_level* L, const double* x, double* y) {
  if (L->has_aff) spmv(&L->A_ff, x, y);
  else {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < L->n_f; i++) y[i] = L->aff_diag[i] * x[i];
  }
}

/* f_smooths (src/FC_Smooth.F90:499-568) */
static void f_smooths(oc_level* L, const double* b, double* x, int first, int its) {
  gather(L->n_f, L->F, b, L->t4);
  if (first) { gather(L->n_f, L->F, x, L->xf); gather(L->n_c, L->C, x, L->xc); }
  spmv(&L->A_fc, L->xc, L->t2);
  axpy(L->n_f, -1.0, L->t2, L->t4);
  for (int it = 0; it < its; it++) {
    aff_mult(L, L->xf, L->t3);
    aypx(L->n_f, -1.0, L->t4, L->t3);
    apply_inv(&L->inv_ff, L->has_aff ? &L->A_ff : NULL, L->aff_diag, L->n_f, L->dff, L->m1, L->m2, L->m3, L->mrhs, L->t3, L->t2);
    axpy(L->n_f, 1.0, L->t2, L->xf);
  }
  scatter(L->n_f, L->F, L->xf, x);
}

/* c_smooths (src/FC_Smooth.F90:572-640) */
static void c_smooths(oc_level* L, const double* b, double* x, int first, int its) {
  gather(L->n_c, L->C, b, L->c4);
  if (first) { gather(L->n_f, L->F, x, L->xf); gather(L->n_c, L->C, x, L->xc); }
  spmv(&L->A_cf, L->xf, L->c2);
  axpy(L->n_c, -1.0, L->c2, L->c4);
  for (int it = 0; it < its; it++) {
    spmv(&L->A_cc, L->xc, L->c3);
    aypx(L->n_c, -1.0, L->c4, L->c3);
    apply_inv(&L->inv_cc, &L->A_cc, NULL, L->n_c, L->dcc, L->m1, L->m2, L->m3, L->mrhs, L->c3, L->c2);
    axpy(L->n_c, 1.0, L->c2, L->xc);
  }
  scatter(L->n_c, L->C, L->xc, x);
}

/* one AIR V-cycle: x = M b (natural ordering of level 1) */
void oc_vcycle(oc_hier* h, const double* b, double* x) {
  const int nl = h->nlev;
  const double* bl = b;
  /* DOWN: b_{l+1} = R b_l, R = [Z I] */
  for (int l = 0; l < nl - 1; l++) {
    oc_level* L = &h->L[l];
    double* bn = (l + 1 < nl - 1) ? h->L[l + 1].b : h->bc;
    gather(L->n_f, L->F, bl, L->t2);
    spmv(&L->Z, L->t2, L->c2);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < L->n_c; j++) bn[j] = bl[L->C[j]] + L->c2[j];
    bl = bn;
  }
  /* COARSE */
  {
    const int64_t n = h->A_c.m;
    const double* bc = (nl > 1) ? h->bc : b;
    double* xc = (nl > 1) ? h->xc : x;
    if (h->rich_its <= 0)
      apply_inv(&h->inv_c, &h->A_c, NULL, n, h->dc, h->m1, h->m2, h->m3, h->mrhs, bc, xc);
    else {
      for (int64_t i = 0; i < n; i++) xc[i] = 0.0;
      for (int it = 0; it < h->rich_its; it++) {
        const double* r = bc;
        if (it > 0) { spmv(&h->A_c, xc, h->rc); aypx(n, -1.0, bc, h->rc); r = h->rc; }
        double* z = (double*)malloc(sizeof(double) * (size_t)n);
        apply_inv(&h->inv_c, &h->A_c, NULL, n, h->dc, h->m1, h->m2, h->m3, h->mrhs, r, z);
        axpy(n, 1.0, z, xc);
        free(z);
      }
    }
  }
  /* UP: x_l = P x_{l+1}, P = [W; I]; then FC relaxation */
  for (int l = nl - 2; l >= 0; l--) {
    oc_level* L = &h->L[l];
    const double* xcn = (l + 1 < nl - 1) ? h->L[l + 1].x : h->xc;
    double* xl = (l == 0) ? x : L->x;
    const double* bl2 = (l == 0) ? b : L->b;
    if (L->one_point) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < L->n_f; i++) xl[L->F[i]] = L->wcol[i] >= 0 ? 1.0 * xcn[L->wcol[i]] : 0.0;
    } else {
      spmv(&L->W, xcn, L->t2);
      scatter(L->n_f, L->F, L->t2, xl);
    }
    scatter(L->n_c, L->C, xcn, xl);
    int first = 1;
    for (int i = 0; i < 10; i++) {
      int s = L->smooth_order[i];
      if (s == 0) break;
      if (s > 0) f_smooths(L, bl2, xl, first, s); else c_smooths(L, bl2, xl, first, -s);
      first = 0;
    }
  }
}

/* ---------------------------------------------------------------------- outer Krylov */
typedef struct { double bnorm, rnorm0, ttol; } conv_t;
static int converged_default(conv_t* st, int it, double rnorm, double rtol, double atol, double dtol) {
  if (it == 0) {
    double snorm = st->bnorm;
    if (snorm == 0.0) snorm = rnorm;
    st->rnorm0 = snorm;
    st->ttol = fmax(rtol * snorm, atol);
  }
  if (isnan(rnorm) || isinf(rnorm)) return -9;
  if (rnorm <= st->ttol) return rnorm < atol ? 3 : 2;
  if (rnorm >= dtol * st->rnorm0) return -4;
  return 0;
}

static void pc_apply(oc_hier* h, const double* r, double* z, int64_t n) {
  if (h) oc_vcycle(h, r, z); else vcopy(n, r, z);
}

/* type 0 richardson / 1 gmres; side 0 left / 1 right; norm 0 preconditioned / 1 unpreconditioned */
int oc_ksp_solve(oc_hier* h, const oc_csr* A, int type, int side, int norm, double rtol, double atol, double dtol,
                 int max_it, int restart, const double* b, double* x, int* its_out, int* reason_out, double* rnorm_out,
                 double* hist, int hist_cap, int* hist_len) {
  const int64_t n = A->m;
  int its = 0, reason = 0, nh = 0;
  double rnorm = 0.0;
  conv_t st; memset(&st, 0, sizeof(st));
  double* r = dalloc(n); double* z = dalloc(n);
#define HIST(v) do { if (hist && nh < hist_cap) hist[nh] = (v); nh++; } while (0)
  if (type == 0) {
    const int prec = (norm == 0);
    double bn = sqrt(dot(n, b, b));
    if (prec && bn != 0.0) { pc_apply(h, b, z, n); bn = sqrt(dot(n, z, z)); }
    st.bnorm = bn;
    spmv(A, x, r); aypx(n, -1.0, b, r);
    for (int i = 0; i < max_it; i++) {
      if (prec) { pc_apply(h, r, z, n); rnorm = sqrt(dot(n, z, z)); } else rnorm = sqrt(dot(n, r, r));
      HIST(rnorm);
      reason = converged_default(&st, i, rnorm, rtol, atol, dtol);
      if (reason) break;
      if (!prec) pc_apply(h, r, z, n);
      axpy(n, 1.0, z, x);
      its++;
      spmv(A, x, r); aypx(n, -1.0, b, r);
    }
    if (!reason) {
      if (prec) { pc_apply(h, r, z, n); rnorm = sqrt(dot(n, z, z)); } else rnorm = sqrt(dot(n, r, r));
      HIST(rnorm);
      if (its >= max_it) { reason = converged_default(&st, its, rnorm, rtol, atol, dtol); if (!reason) reason = -3; }
    }
  } else {
    const int m = restart;
    const int right = (side == 1);
    double* V = dalloc((int64_t)(m + 1) * n);
    double* w2 = dalloc(n);
    double* hh = (double*)calloc((size_t)(m + 2) * (m + 1), sizeof(double));
    double *cs = (double*)calloc(m + 1, 8), *sn = (double*)calloc(m + 1, 8), *grs = (double*)calloc(m + 2, 8);
    double *lhh = (double*)calloc(m + 1, 8), *nrs = (double*)calloc(m + 1, 8);
#define H(i, j) hh[(size_t)(j) * (m + 2) + (i)]
    double bn = sqrt(dot(n, b, b));
    if (!right && bn != 0.0) { pc_apply(h, b, z, n); bn = sqrt(dot(n, z, z)); }
    st.bnorm = bn;
    const double haptol = 1e-30;
    int first = 1;
    while (!reason) {
      double* V0 = V;
      if (right) { spmv(A, x, V0); aypx(n, -1.0, b, V0); }
      else { spmv(A, x, r); aypx(n, -1.0, b, r); pc_apply(h, r, V0, n); }
      double res = sqrt(dot(n, V0, V0));
      rnorm = res;
      if (first) {
        HIST(res);
        if (res == 0.0) { reason = 3; break; }
        reason = converged_default(&st, its, res, rtol, atol, dtol);
        first = 0;
        if (reason) break;
      }
      axpby(n, 0.0, V0, 1.0 / res, V0);
      memset(hh, 0, sizeof(double) * (size_t)(m + 2) * (m + 1));
      grs[0] = res;
      int it = 0, hapend = 0;
      while (!reason && it < m && its < max_it) {
        double* vi = V + (int64_t)it * n; double* w = V + (int64_t)(it + 1) * n;
        if (right) { pc_apply(h, vi, z, n); spmv(A, z, w); } else { spmv(A, vi, z); pc_apply(h, z, w, n); }
        for (int j = 0; j <= it; j++) lhh[j] = dot(n, w, V + (int64_t)j * n);
        for (int j = 0; j <= it; j++) { H(j, it) = lhh[j]; axpy(n, -lhh[j], V + (int64_t)j * n, w); }
        double tt = sqrt(dot(n, w, w));
        if (tt != 0.0) axpby(n, 0.0, w, 1.0 / tt, w);
        H(it + 1, it) = tt;
        double hapbnd = grs[it] != 0.0 ? fabs(tt / grs[it]) : haptol;
        if (hapbnd > haptol) hapbnd = haptol;
        if (tt < hapbnd) hapend = 1;
        for (int j = 0; j < it; j++) {
          double t1 = H(j, it);
          H(j, it) = cs[j] * t1 + sn[j] * H(j + 1, it);
          H(j + 1, it) = -sn[j] * t1 + cs[j] * H(j + 1, it);
        }
        if (!hapend) {
          double t = sqrt(H(it, it) * H(it, it) + H(it + 1, it) * H(it + 1, it));
          if (t == 0.0) { reason = -5; break; }
          cs[it] = H(it, it) / t; sn[it] = H(it + 1, it) / t;
          grs[it + 1] = -sn[it] * grs[it];
          grs[it] = cs[it] * grs[it];
          H(it, it) = cs[it] * H(it, it) + sn[it] * H(it + 1, it);
          res = fabs(grs[it + 1]);
        } else res = 0.0;
        it++; its++;
        rnorm = res;
        HIST(res);
        reason = converged_default(&st, its, res, rtol, atol, dtol);
        if (hapend) { if (!reason) reason = -5; break; }
      }
      if (it > 0) {
        int k = it - 1;
        for (int j = 0; j <= m; j++) nrs[j] = 0.0;
        if (H(k, k) != 0.0) {
          nrs[k] = grs[k] / H(k, k);
          for (int ii = 1; ii <= k; ii++) {
            int kk = k - ii;
            double tsum = grs[kk];
            for (int j = kk + 1; j <= k; j++) tsum -= H(kk, j) * nrs[j];
            nrs[kk] = tsum / H(kk, kk);
          }
        }
        if (right) {
          for (int64_t i = 0; i < n; i++) w2[i] = 0.0;
          for (int j = 0; j < it; j++) axpy(n, nrs[j], V + (int64_t)j * n, w2);
          pc_apply(h, w2, z, n);
          axpy(n, 1.0, z, x);
        } else for (int j = 0; j < it; j++) axpy(n, nrs[j], V + (int64_t)j * n, x);
      }
      if (!reason && its >= max_it) reason = -3;
    }
    free(V); free(w2); free(hh); free(cs); free(sn); free(grs); free(lhh); free(nrs);
#undef H
  }
#undef HIST
  free(r); free(z);
  if (its_out) *its_out = its;
  if (reason_out) *reason_out = reason;
  if (rnorm_out) *rnorm_out = rnorm;
  if (hist_len) *hist_len = nh;
  return 0;
}

/* y = A x, exposed for SpMV parity / bandwidth baselines */
void oc_spmv(const oc_csr* A, const double* x, double* y) { spmv(A, x, y); }
