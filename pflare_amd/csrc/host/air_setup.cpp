// PCAIR hierarchy construction: the level loop of setup_air_pcmg
// (src/AIR_MG_Setup.F90:138-962) with the operator construction of
// src/AIR_Operators_Setup.F90:36-1085, default AIRG path (z_type product, PMISR+DDC,
// no reuse, no processor agglomeration, no near-nullspace constraints).
#include "air.hpp"
#include "dense.hpp"
#include <cstdio>

namespace airhip {

int AirOptions::set_smooth_type(const char* s) {
  // src/PCAIR_Interfaces.F90:1821-1882: "ffc" -> (2,-1,0,...)
  int out[10] = {0}; int n = 0;
  for (const char* p = s; *p; p++) {
    int val;
    if (*p == 'f' || *p == 'F') val = 1; else if (*p == 'c' || *p == 'C') val = -1; else return -1;
    if (n > 0 && ((out[n - 1] > 0) == (val > 0))) out[n - 1] += val;
    else { if (n == 10) return -1; out[n++] = val; }
  }
  if (n == 0) return -1;
  for (int i = 0; i < 10; i++) smooth_order[i] = out[i];
  return 0;
}

// one-point classical prolongator (src/Grid_Transfer.F90:94-220)
static void one_point_W(const Csr& A_fc, Arr<int32_t>& wcol) {
  wcol.alloc(A_fc.m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A_fc.m; i++) {
    double best = -1.0; int32_t col = -1;
    for (int64_t k = A_fc.rp[i]; k < A_fc.rp[i + 1]; k++) {
      double a = std::fabs(A_fc.v[k]);
      if (a > best) { best = a; col = A_fc.ci[k]; }
    }
    wcol[i] = col;
  }
}

// R = [Z I] in level ordering (src/Grid_Transfer.F90:588-815)
static Csr build_R(const Csr& Z, const Arr<int32_t>& F, const Arr<int32_t>& C, int64_t n) {
  const int64_t nc = C.size();
  Csr R; R.m = nc; R.n = n; R.rp.alloc(nc + 1);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j <= nc; j++) R.rp[j] = (j < nc ? Z.rp[j] : Z.rp[nc]) + j;
  R.ci.alloc(R.rp[nc]); R.v.alloc(R.rp[nc]);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < nc; j++) {
    int64_t o = R.rp[j]; const int32_t cj = C[j]; bool placed = false;
    for (int64_t k = Z.rp[j]; k < Z.rp[j + 1]; k++) {
      int32_t col = F[Z.ci[k]];
      if (!placed && col > cj) { R.ci[o] = cj; R.v[o] = 1.0; o++; placed = true; }
      R.ci[o] = col; R.v[o] = Z.v[k]; o++;
    }
    if (!placed) { R.ci[o] = cj; R.v[o] = 1.0; o++; }
  }
  return R;
}

// P = [W; I] in level ordering (src/Grid_Transfer.F90:329-461)
static Csr build_P(const Level& L) {
  const int64_t n = L.n, nc = L.n_c;
  Csr P; P.m = n; P.n = nc;
  Arr<int64_t> cnt(n);
  Arr<int32_t> where(n);   // >=0: index into F ; <0: -(index into C)-1
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < L.n_f; i++) where[L.F[i]] = (int32_t)i;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < nc; j++) where[L.C[j]] = -(int32_t)j - 1;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; r++) {
    int32_t w = where[r];
    if (w < 0) cnt[r] = 1;
    else cnt[r] = L.one_point ? (L.wcol[w] >= 0 ? 1 : 0) : (L.W.rp[w + 1] - L.W.rp[w]);
  }
  P.rp.alloc(n + 1);
  exclusive_scan(cnt.data(), n, P.rp.data());
  P.ci.alloc(P.rp[n]); P.v.alloc(P.rp[n]);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; r++) {
    int32_t w = where[r]; int64_t o = P.rp[r];
    if (w < 0) { P.ci[o] = -w - 1; P.v[o] = 1.0; }
    else if (L.one_point) { if (L.wcol[w] >= 0) { P.ci[o] = L.wcol[w]; P.v[o] = 1.0; } }
    else for (int64_t k = L.W.rp[w]; k < L.W.rp[w + 1]; k++) { P.ci[o] = L.W.ci[k]; P.v[o] = L.W.v[k]; o++; }
  }
  return P;
}

static Csr transpose_csr(const Csr& A) {
  Csr T; T.m = A.n; T.n = A.m;
  T.rp.alloc(T.m + 1);
  for (int64_t i = 0; i <= T.m; i++) T.rp[i] = 0;
  for (int64_t k = 0; k < A.nnz(); k++) T.rp[A.ci[k] + 1]++;
  for (int64_t i = 0; i < T.m; i++) T.rp[i + 1] += T.rp[i];
  T.ci.alloc(A.nnz()); T.v.alloc(A.nnz());
  Arr<int64_t> next(T.m + 1);
  for (int64_t i = 0; i <= T.m; i++) next[i] = T.rp[i];
  for (int64_t r = 0; r < A.m; r++)           // rows ascending -> columns of every transposed row ascending
    for (int64_t k = A.rp[r]; k < A.rp[r + 1]; k++) { const int64_t q = next[A.ci[k]]++; T.ci[q] = (int32_t)r; T.v[q] = A.v[k]; }
  return T;
}

// Richardson improvement of the ideal restrictor on its own pattern (improve_z, src/Grid_Transfer_Improve.F90:200-371):
//   Z <- Z - ((Z A_ff + A_cf) diag(M)) restricted to pattern(Z),  dm = diagonal of the A_ff inverse.
// A_ff == nullptr: MATDIAGONAL operator with diagonal aff_d.
static void improve_z(Csr& Z, const Csr* A_ff, const double* aff_d, const Csr& A_cf, const double* dm, int its) {
  for (int it = 0; it < its; it++) {
#pragma omp parallel
    {
      std::vector<double> t;
#pragma omp for schedule(dynamic, 256)
      for (int64_t i = 0; i < Z.m; i++) {
        const int64_t p0 = Z.rp[i], nz = Z.rp[i + 1] - p0;
        if (nz == 0) continue;
        const int32_t* J = Z.ci.data() + p0;
        t.assign(nz, 0.0);
        auto at = [&](int32_t c) -> int64_t { const int32_t* q = std::lower_bound(J, J + nz, c); return (q != J + nz && *q == c) ? q - J : -1; };
        for (int64_t a = 0; a < nz; a++) {
          const double z = Z.v[p0 + a];
          if (A_ff) {
            for (int64_t k = A_ff->rp[J[a]]; k < A_ff->rp[J[a] + 1]; k++) { const int64_t q = at(A_ff->ci[k]); if (q >= 0) t[q] += z * A_ff->v[k]; }
          } else t[a] += z * aff_d[J[a]];
        }
        for (int64_t k = A_cf.rp[i]; k < A_cf.rp[i + 1]; k++) { const int64_t q = at(A_cf.ci[k]); if (q >= 0) t[q] += A_cf.v[k]; }
        for (int64_t a = 0; a < nz; a++) Z.v[p0 + a] -= t[a] * dm[J[a]];
      }
    }
  }
}

// improve_w (src/Grid_Transfer_Improve.F90:21-196): W <- W - (diag(M) (A_ff W + A_fc)) restricted to pattern(W)
static void improve_w(Csr& W, const Csr* A_ff, const double* aff_d, const Csr& A_fc, const double* dm, int its) {
  for (int it = 0; it < its; it++) {
    Arr<double> nv(W.nnz());
#pragma omp parallel
    {
      std::vector<double> t;
#pragma omp for schedule(dynamic, 256)
      for (int64_t i = 0; i < W.m; i++) {
        const int64_t p0 = W.rp[i], nz = W.rp[i + 1] - p0;
        if (nz == 0) continue;
        const int32_t* J = W.ci.data() + p0;
        t.assign(nz, 0.0);
        auto at = [&](int32_t c) -> int64_t { const int32_t* q = std::lower_bound(J, J + nz, c); return (q != J + nz && *q == c) ? q - J : -1; };
        auto add_row = [&](int64_t r, double a) {
          for (int64_t k = W.rp[r]; k < W.rp[r + 1]; k++) { const int64_t q = at(W.ci[k]); if (q >= 0) t[q] += a * W.v[k]; }
        };
        if (A_ff) for (int64_t k = A_ff->rp[i]; k < A_ff->rp[i + 1]; k++) add_row(A_ff->ci[k], A_ff->v[k]);
        else add_row(i, aff_d[i]);
        for (int64_t k = A_fc.rp[i]; k < A_fc.rp[i + 1]; k++) { const int64_t q = at(A_fc.ci[k]); if (q >= 0) t[q] += A_fc.v[k]; }
        for (int64_t a = 0; a < nz; a++) nv[p0 + a] = W.v[p0 + a] - dm[i] * t[a];
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < W.nnz(); k++) W.v[k] = nv[k];
  }
}

static void coarse_solver(const Csr& A, const AirOptions& o, InvOp& op, bool final_solver = false) {
  if (o.coarse_lu && final_solver) {
    // -mg_coarse_pc_type lu (tests/Makefile:144-147): the inverse of the coarsest operator by LU, applied as a
    // (dense) assembled operator
    const int64_t n = A.m;
    if (n > 4096) throw std::runtime_error("-mg_coarse_pc_type lu: coarsest operator too large for a dense inverse (> 4096 rows)");
    Dense D((int)n, (int)n);
    for (int64_t i = 0; i < n; i++) for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) D((int)i, A.ci[k]) = A.v[k];
    op = InvOp(); op.kind = InvOp::ASSEMBLED;
    Csr& M = op.mat; M.m = M.n = n;
    M.rp.alloc(n + 1); M.ci.alloc(n * n); M.v.alloc(n * n);
    for (int64_t i = 0; i <= n; i++) M.rp[i] = i * n;
    const Dense X = lu_inverse(D);                            // one factorisation, n right-hand sides
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
      for (int64_t j = 0; j < n; j++) { M.ci[i * n + j] = (int32_t)j; M.v[i * n + j] = X((int)i, (int)j); }
    return;
  }
  OpView v; v.A = &A; v.n = A.m;
  build_inverse(v, o.coarsest_inverse_type, o.coarsest_poly_order, o.coarsest_inverse_sparsity_order,
                o.coarsest_matrix_free_polys, o.coarsest_diag_scale_polys, o.seed, op, nullptr);
}

std::unique_ptr<Hierarchy> setup_air(const Csr& A0, const AirOptions& o, SetupAccel* accel) {
  struct AccelScope { SetupAccel* prev; explicit AccelScope(SetupAccel* a) : prev(g_setup_accel) { g_setup_accel = a; } ~AccelScope() { g_setup_accel = prev; } } accel_scope(accel);
  auto H = std::make_unique<Hierarchy>();
  H->opt = o;
  H->timers.assign(8, 0.0);
  Csr owned;              // level operators below the top are owned here
  const Csr* cur = &A0;
  // with an accelerator the level operators below the top stay on it until a host routine asks for one
  auto on_host = [&]() -> const Csr& { if (!cur->on_host()) accel->materialize(owned); return *cur; };
  const bool any_c = o.any_c_smooths();
  // the recipe in which no host routine reads A_cf, the assembled A_ff inverse or Z: the accelerator may leave them on
  // the device (SetupAccel::defer_downloads); every other combination of options gets all blocks on the host as before
  if (accel) accel->plain_recipe(o.z_type == 0 && o.improve_z_its == 0 && o.improve_w_its == 0 && o.strong_r_threshold == 0.0 && !o.symmetric &&
                                 o.one_point_classical_prolong && !any_c && !o.full_smoothing && !o.diag_scale_polys && !o.matrix_free_polys);
  bool auto_truncated = false;
  const double t_setup = now_s();
  if (o.print_stats_timings) std::printf(" Timers are cumulative\n");     // src/AIR_MG_Setup.F90:130

  for (int our_level = 1; our_level < o.max_levels; our_level++) {
    const int64_t n = cur->m;
    // ---- auto truncation test (AIR_MG_Setup.F90:158-256)
    if (o.auto_truncate_start_level != -1 && our_level >= o.auto_truncate_start_level) {
      double t0 = now_s();
      InvOp cop;
      // the accelerator may evaluate the test (device Arnoldi + Newton apply); its ratio differs from the host's in
      // the last digits, so it only decides when it is a factor 2 away from the tolerance - anything closer, and
      // every solver that is kept, goes through the host code below
      double dev_ratio = 0.0;
      const bool dev_ok = accel && o.coarsest_matrix_free_polys && !o.coarsest_diag_scale_polys &&
                          (int64_t)o.coarsest_poly_order + 1 <= n &&
                          accel->truncation_ratio(*cur, o.coarsest_inverse_type, o.coarsest_poly_order, o.seed, dev_ratio);
      bool decided = dev_ok && dev_ratio > 2.0 * o.auto_truncate_tol;        // clearly not yet: carry on coarsening
      if (!decided) {
        coarse_solver(on_host(), o, cop);
        if (dev_ok && dev_ratio < 0.5 * o.auto_truncate_tol) {                // clearly good enough: truncate here
          auto_truncated = true; H->coarse_inv = std::move(cop); decided = true;
        }
      }
      if (!decided) {
      Arr<double> rnd(n), sol(n), tmp(n);
      const uint64_t key = rng_key(o.seed + 7919);   // distinct stream from the polynomial RHS
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) rnd[i] = uniform01(key, (uint64_t)i);
      OpView v; v.A = cur; v.n = n;
      host_apply_inv(cop, v, rnd.data(), sol.data());
      spmv(*cur, sol.data(), tmp.data());
      const double nb = det_sum(0, n, [&](int64_t i) { return rnd[i] * rnd[i]; });
      const double nr = det_sum(0, n, [&](int64_t i) { const double d = tmp[i] - rnd[i]; return d * d; });
      if (std::sqrt(nr) / std::sqrt(nb) < o.auto_truncate_tol) { auto_truncated = true; H->coarse_inv = std::move(cop); }
      }
      H->timers[7] += now_s() - t0;
    }
    // ---- CF splitting
    Arr<int32_t> markers;
    int64_t n_f = 0, n_c = 0;
    if (!auto_truncated) {
      double t0 = now_s();
      if (!(accel && accel->cf_splitting(*cur, o, markers))) compute_cf_splitting(on_host(), o, markers);
#pragma omp parallel for schedule(static) reduction(+ : n_f)
      for (int64_t i = 0; i < n; i++) n_f += (markers[i] == F_POINT) ? 1 : 0;
      n_c = n - n_f;
      H->timers[0] += now_s() - t0;
    }
    const bool cont = !auto_truncated && (n_c > o.coarse_eq_limit && n_f != 0);
    if (!cont) break;

    auto Lp = std::make_unique<Level>();
    Level& L = *Lp;
    L.n = n; L.n_f = n_f; L.n_c = n_c; L.nnz_A = cur->nnz();
    // index sets + position-within-set map
    double t0 = now_s();
    Arr<int32_t> newidx(n);
    {
      Arr<int64_t> isf(n), pre(n + 1);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) isf[i] = markers[i] == F_POINT ? 1 : 0;
      exclusive_scan(isf.data(), n, pre.data());
      L.F.alloc(n_f); L.C.alloc(n_c);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) {
        if (markers[i] == F_POINT) { newidx[i] = (int32_t)pre[i]; L.F[pre[i]] = (int32_t)i; }
        else { newidx[i] = (int32_t)(i - pre[i]); L.C[i - pre[i]] = (int32_t)i; }
      }
    }
    if (!(accel && accel->split_fc(*cur, markers, newidx, n_f, n_c, L.A_ff, L.A_fc, L.A_cf, any_c ? &L.A_cc : nullptr)))
      split_fc(on_host(), markers.data(), newidx.data(), L.F, L.C, L.A_ff, L.A_fc, L.A_cf, any_c ? &L.A_cc : nullptr);
    H->timers[1] += now_s() - t0;

    for (int i = 0; i < 10; i++) L.smooth_order[i] = o.smooth_order[i];
    int inv_type_aff = o.inverse_type, sparsity_aff = o.inverse_sparsity_order;
    // diagonal A_ff detection (AIR_MG_Setup.F90:456-522)
    L.aff_diag = (o.strong_threshold == 0.0) ? true : is_diag_only(L.A_ff);
    if (L.aff_diag) {
      L.aff_d.alloc(n_f);
      get_diagonal(L.A_ff, L.aff_d.data());
      L.A_ff = Csr();
      sparsity_aff = 0;
      if (inv_type_aff != INV_WJACOBI && o.poly_order > 2)
        for (int i = 0; i < 10; i++) if (L.smooth_order[i] > 0) L.smooth_order[i] = 1;
    }
    // ---- strong R threshold: Z (and an ideal W) are built from a copy of A with the small entries of every
    // row removed, diagonal kept (src/AIR_Operators_Setup.F90:46-100, 219-276)
    Csr Aff_drop, Afc_drop, Acf_drop;
    const bool r_dropped = o.strong_r_threshold != 0.0;
    if (r_dropped) {
      t0 = now_s();
      Csr Ad = remove_small(on_host(), o.strong_r_threshold, 1, false, 0);
      H->timers[6] += now_s() - t0;
      t0 = now_s();
      split_fc(Ad, markers.data(), newidx.data(), L.F, L.C, Aff_drop, Afc_drop, Acf_drop, nullptr);
      H->timers[1] += now_s() - t0;
    }
    const Csr& Acf_z = r_dropped ? Acf_drop : L.A_cf;
    const Csr& Afc_w = r_dropped ? Afc_drop : L.A_fc;
    // ---- A_ff^-1 for smoothing and for the grid transfers
    t0 = now_s();
    OpView aff; aff.n = n_f;
    if (L.aff_diag) aff.diag = L.aff_d.data(); else aff.A = &L.A_ff;
    InvOp inv_asm;
    const bool product_z = o.z_type == 0;
    const bool full = o.full_smoothing;
    if (full && (o.improve_z_its || o.improve_w_its))
      throw std::runtime_error("improve_z / improve_w are not available with full up / down smoothing");
    // the smoother: A_ff^-1, or with full smoothing an inverse of the whole level operator (AIR_Operators_Setup.F90:113-119)
    OpView sm = aff;
    if (full) { sm = OpView(); sm.A = &on_host(); sm.n = n; }
    build_inverse(sm, inv_type_aff, o.poly_order, sparsity_aff, o.matrix_free_polys, o.diag_scale_polys,
                  o.seed, L.inv_ff, (product_z && !r_dropped && !full) ? &inv_asm : nullptr);
    if (product_z && (r_dropped || full)) {
      // a second, always assembled, inverse of the (dropped) A_ff for the grid transfers (AIR_Operators_Setup.F90:413-438)
      OpView affd = aff;
      if (r_dropped) { affd = OpView(); affd.n = n_f; affd.A = &Aff_drop; }
      build_inverse(affd, inv_type_aff, o.poly_order, sparsity_aff, false, o.diag_scale_polys, o.seed, inv_asm, nullptr);
    }
    const InvOp& invZ = (inv_asm.kind != InvOp::NONE) ? inv_asm : L.inv_ff;
    if (product_z && invZ.kind != InvOp::ASSEMBLED && invZ.kind != InvOp::DIAG)
      throw std::runtime_error("grid transfers need an assembled A_ff inverse");
    if (any_c && !full) {
      OpView acc; acc.A = &L.A_cc; acc.n = n_c;
      build_inverse(acc, o.c_inverse_type, o.c_poly_order, o.c_inverse_sparsity_order, o.matrix_free_polys,
                    o.diag_scale_polys, o.seed, L.inv_cc, nullptr);
    }
    H->timers[2] += now_s() - t0;

    // diagonal of the smoother's A_ff inverse (improve_z / improve_w use only that, Grid_Transfer_Improve.F90:235-241)
    Arr<double> inv_diag_store;
    auto inv_diag = [&]() -> Arr<double>& {
      if (inv_diag_store.size() == 0 && n_f > 0) {
        inv_diag_store.alloc(n_f);
        if (L.inv_ff.kind == InvOp::ASSEMBLED) get_diagonal(L.inv_ff.mat, inv_diag_store.data());
        else if (L.inv_ff.kind == InvOp::DIAG) { for (int64_t i = 0; i < n_f; i++) inv_diag_store[i] = L.inv_ff.d[i]; }
        else throw std::runtime_error("improve_z / improve_w need an assembled A_ff inverse (not matrix-free)");
      }
      return inv_diag_store;
    };
    // ---- W / P
    t0 = now_s();
    Csr Mz_store;
    const Csr* Mz = nullptr;
    if (product_z) {
      if (invZ.kind == InvOp::ASSEMBLED) Mz = &invZ.mat;
      else { Mz_store = csr_from_diag(invZ.d.data(), n_f); Mz = &Mz_store; }
    }
    if (o.symmetric) {
      // -pc_air_symmetric: P = R^T, i.e. W = Z^T, set once Z exists (AIR_Operators_Setup.F90:898-902)
      L.one_point = false;
    } else if (o.one_point_classical_prolong) {
      L.one_point = true;
      if (!(accel && accel->one_point_w(L.A_fc, L.wcol))) {
        if (accel) accel->materialize(L.A_fc);     // (a stub when its download was deferred)
        one_point_W(L.A_fc, L.wcol);
      }
    } else {
      if (!product_z) throw std::runtime_error("lAIR / SAI restrictors need the one-point classical prolongator");
      L.one_point = false;
      SpgemmDrop d; d.enabled = true; d.tol = o.r_drop; d.negate = true;
      if (o.improve_w_its == 0) L.W = spgemm(*Mz, Afc_w, d);
      else {
        d.enabled = false;
        Csr Wf = spgemm(*Mz, Afc_w, d);
        improve_w(Wf, L.aff_diag ? nullptr : &L.A_ff, L.aff_d.data(), L.A_fc, inv_diag().data(), o.improve_w_its);
        L.W = remove_small(Wf, o.r_drop, 1, false, 0);
      }
    }
    H->timers[3] += now_s() - t0;
    // ---- Z / R
    t0 = now_s();
    Csr Zfull;
    if (product_z) {
      SpgemmDrop d; d.enabled = true; d.tol = o.r_drop; d.negate = true;
      if (o.improve_z_its == 0) L.Z = accel ? accel->spgemm_z(Acf_z, *Mz, d) : spgemm(Acf_z, *Mz, d);
      else { d.enabled = false; Zfull = spgemm(Acf_z, *Mz, d); }
    } else {
      // lAIR / SAI (AIR_Operators_Setup.F90:698-778): the F neighbourhood of every C point out to
      // lair_distance comes from the dropped operators, the values from the full ones
      Csr Aff_full_store;
      const Csr* Aff_full = &L.A_ff;
      if (L.aff_diag) { Aff_full_store = csr_from_diag(L.aff_d.data(), n_f); Aff_full = &Aff_full_store; }
      const Csr& Aff_pat = r_dropped ? Aff_drop : *Aff_full;
      Csr S_store;
      const Csr* S = &Acf_z;
      if (o.lair_distance >= 2) {
        Csr Pw;
        const Csr* pw = &Aff_pat;
        for (int order = 3; order <= o.lair_distance; order++) { Pw = spgemm(Aff_pat, *pw); pw = &Pw; }
        S_store = spgemm(Acf_z, *pw);
        S = &S_store;
      }
      Zfull = sai_z(*Aff_full, &L.A_cf, *S, o.z_type == 1);
    }
    if (!product_z || o.improve_z_its != 0) {
      // improve_z runs on the undropped Z with the full A_ff / A_cf and the smoother's inverse
      // (AIR_Operators_Setup.F90:819-830), then the relative row drop
      if (o.improve_z_its != 0)
        improve_z(Zfull, L.aff_diag ? nullptr : &L.A_ff, L.aff_d.data(), L.A_cf, inv_diag().data(), o.improve_z_its);
      L.Z = remove_small(Zfull, o.r_drop, 1, false, 0);
      Zfull = Csr();
    }
    Mz_store = Csr();
    inv_asm = InvOp();
    H->timers[4] += now_s() - t0;
    if (o.symmetric) L.W = transpose_csr(L.Z);

    // ---- coarse operator R (A P) with drop / lump (AIR_Operators_Setup.F90:960-1085)
    t0 = now_s();
    Csr coarse;
    SpgemmDrop d_rap; d_rap.enabled = true; d_rap.tol = o.a_drop; d_rap.lump = o.a_lump;
    // with the plain Z product and the one-point prolongator the accelerator assembles R = [Z I] and P = [W; I] itself,
    // where Z and the level operator already are (reference device twin: src/Grid_Transfer.kokkos.cxx)
    if (accel && L.one_point && !o.symmetric && accel->rap_from_z(L.Z, L.F, L.C, L.wcol, *cur, d_rap, coarse)) {
      H->timers[5] += now_s() - t0;
    } else {
      if (accel) accel->materialize(L.Z);        // (a stub when its download was deferred)
      Csr P = build_P(L);
      Csr R = build_R(L.Z, L.F, L.C, n);
      SpgemmDrop d = d_rap;
      if (accel) coarse = accel->rap(R, *cur, P, d, true);
      else {
        Csr AP = spgemm(*cur, P);
        coarse = spgemm(R, AP, d);      // drop / lump fused into the product (timer: RAP)
      }
      H->timers[5] += now_s() - t0;
    }
    if (!any_c) { /* A_cf is only needed again for C smoothing; keep it anyway for export */ }
    if (o.print_stats_timings)
      std::printf(" ~~~~~~~~~~~~ Level %d\n Global rows %lld Global F-points %lld Global C-points %lld  (nnz %lld, A_ff %s, t=%.2fs, outside the timers %.2fs)\n",
                  our_level, (long long)n, (long long)n_f, (long long)n_c, (long long)L.nnz_A,
                  L.aff_diag ? "diagonal" : "sparse", now_s() - t_setup,
                  now_s() - t_setup - (H->timers[0] + H->timers[1] + H->timers[2] + H->timers[3] + H->timers[4] + H->timers[5] + H->timers[6] + H->timers[7]));
    if (o.print_stats_timings)   // labels of print_timers() (src/Timers.F90:30-46): tools/parse_pflare_output.py reads them
      std::printf(" coarsen time     : %.6f\n extract time     : %.6f\n proc agglom time : %.6f\n inverse time     : %.6f\n"
                  " restrict time    : %.6f\n prolong time     : %.6f\n constrain time   : %.6f\n rap time         : %.6f\n"
                  " identity time    : %.6f\n drop time        : %.6f\n truncate time    : %.6f\n",
                  H->timers[0], H->timers[1], 0.0, H->timers[2], H->timers[4], H->timers[3], 0.0, H->timers[5], 0.0, H->timers[6],
                  H->timers[7]);
    if (o.print_stats_timings) std::fflush(stdout);
    if (full) {      // the cycle needs the level operator itself
      L.full = true;
      if (cur == &owned) L.A_full = std::move(owned); else L.A_full = A0.clone();
    }
    H->levels.push_back(std::move(Lp));
    owned = std::move(coarse);
    cur = &owned;
  }

  // ---- coarse grid solver (AIR_MG_Setup.F90:933-962, 1110-1136)
  if (cur == &A0) H->coarse_A = A0.clone(); else { on_host(); H->coarse_A = std::move(owned); }
  if (!auto_truncated) {
    double t0 = now_s();
    if (H->levels.empty() && !o.coarse_lu) {
      // "Only a single level, defaulting to Jacobi PC" (AIR_MG_Setup.F90:1167-1174)
      H->single_level_jacobi = true;
      const int64_t n = H->coarse_A.m;
      H->coarse_inv.kind = InvOp::DIAG; H->coarse_inv.d.alloc(n);
      get_diagonal(H->coarse_A, H->coarse_inv.d.data());
      for (int64_t i = 0; i < n; i++) H->coarse_inv.d[i] = 1.0 / H->coarse_inv.d[i];
    } else coarse_solver(H->coarse_A, o, H->coarse_inv, true);
    H->timers[2] += now_s() - t0;
  } else if (o.coarse_lu) coarse_solver(H->coarse_A, o, H->coarse_inv, true);   // PETSc's option overrides the shell PC either way
  if (o.print_stats_timings)
    std::printf(" ~~~~~~~~~~~~ Coarse grid %d\n Global rows %lld\n ~~~~~~~~~~~~\n Total cumulative setup time : %.3f\n ~~~~~~~~~~~~\n",
                H->no_levels(), (long long)H->coarse_A.m, now_s() - t_setup);
  return H;
}

}  // namespace airhip
