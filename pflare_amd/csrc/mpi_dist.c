/* MPI host layer over libairhip (include/airhip_mpi.h): PCAIR on the ranks of an MPI communicator, one rank per
 * GPU, contiguous row blocks - the reference's MPIAIJ model (src/PCAIR.c:182-201 under mpiexec; rank-owned rows with
 * off-process columns fetched in front of each product, src/PMISR_Module.F90:174-180; coarse rows stay with their
 * parent C point, src/Grid_Transfer.F90:745-746).
 *
 * Setup: rank 0 gathers the operator, builds the hierarchy with the library's (bit-reproducible) setup, cuts it with
 * airhip_partition_* and sends every rank its piece as a sequence of named flat arrays; every rank uploads its piece
 * with airhip_pcair_load_distributed.  Transports of the halo exchange inside the cycle: MPI_Isend / MPI_Irecv on
 * pinned host buffers (callbacks of airhip_comm_create_staged) or RCCL (unique id broadcast with MPI_Bcast).
 *
 * Plain C + MPI; no PETSc.  Built by `make -C pflare_amd/csrc mpi` into pflare_amd/libairhip_mpi.so.
 */
#include "airhip_mpi.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[512] = "";
const char* airhip_mpi_last_error(void) { return g_err; }

#define FAIL(code, ...)                         \
  do {                                          \
    snprintf(g_err, sizeof g_err, __VA_ARGS__); \
    rc = (code);                                \
    goto done;                                  \
  } while (0)
#define AIR(expr)                                                                  \
  do {                                                                             \
    int _r = (expr);                                                               \
    if (_r != AIRHIP_OK) FAIL(_r, "%s: %s", #expr, airhip_last_error());           \
  } while (0)
#define MPIC(expr)                                                                 \
  do {                                                                             \
    if ((expr) != MPI_SUCCESS) FAIL(AIRHIP_ERR_INTERNAL, "MPI call failed: %s", #expr); \
  } while (0)

struct airhip_mpi_pcair_s {
  MPI_Comm comm;              /* duplicate of the caller's communicator: halo traffic never meets the caller's tags */
  int rank, size;
  airhip_pcair pc;
  airhip_comm tr;
  int64_t row_start, m_local, n_global;
  int cb_error;               /* set by a callback that failed */
};

/* messages larger than INT_MAX elements are split */
#define CHUNK ((int64_t)1 << 28)
static int send_big(const void* buf, int64_t bytes, int dst, int tag, MPI_Comm comm) {
  const char* p = (const char*)buf;
  while (bytes > 0) {
    const int64_t n = bytes < CHUNK ? bytes : CHUNK;
    if (MPI_Send(p, (int)n, MPI_BYTE, dst, tag, comm) != MPI_SUCCESS) return 1;
    p += n; bytes -= n;
  }
  return 0;
}
static int recv_big(void* buf, int64_t bytes, int src, int tag, MPI_Comm comm) {
  char* p = (char*)buf;
  while (bytes > 0) {
    const int64_t n = bytes < CHUNK ? bytes : CHUNK;
    if (MPI_Recv(p, (int)n, MPI_BYTE, src, tag, comm, MPI_STATUS_IGNORE) != MPI_SUCCESS) return 1;
    p += n; bytes -= n;
  }
  return 0;
}

/* ---- transport callbacks (airhip_exchange_fn / airhip_allreduce_fn): host buffers, peer after peer */
static int mpi_exchange(void* user, int npeers, const int* peers, const double* send, const int64_t* send_off, double* recv,
                        const int64_t* recv_off) {
  struct airhip_mpi_pcair_s* h = (struct airhip_mpi_pcair_s*)user;
  MPI_Request stack_req[64];
  MPI_Request* req = 2 * npeers <= 64 ? stack_req : (MPI_Request*)malloc(sizeof(MPI_Request) * 2 * (size_t)npeers);
  int nreq = 0, bad = 0;
  for (int k = 0; k < npeers && !bad; k++) {
    const int64_t nr = recv_off[k + 1] - recv_off[k];
    if (nr > 0x7fffffffll) bad = 1;
    else if (nr) bad |= MPI_Irecv(recv + recv_off[k], (int)nr, MPI_DOUBLE, peers[k], 77, h->comm, &req[nreq++]) != MPI_SUCCESS;
  }
  for (int k = 0; k < npeers && !bad; k++) {
    const int64_t ns = send_off[k + 1] - send_off[k];
    if (ns > 0x7fffffffll) bad = 1;
    else if (ns) bad |= MPI_Isend(send + send_off[k], (int)ns, MPI_DOUBLE, peers[k], 77, h->comm, &req[nreq++]) != MPI_SUCCESS;
  }
  if (nreq) bad |= MPI_Waitall(nreq, req, MPI_STATUSES_IGNORE) != MPI_SUCCESS;
  if (req != stack_req) free(req);
  if (bad) h->cb_error = 1;
  return bad;
}
static int mpi_allreduce(void* user, double* vals, int n) {
  struct airhip_mpi_pcair_s* h = (struct airhip_mpi_pcair_s*)user;
  const int bad = MPI_Allreduce(MPI_IN_PLACE, vals, n, MPI_DOUBLE, MPI_SUM, h->comm) != MPI_SUCCESS;
  if (bad) h->cb_error = 1;
  return bad;
}

/* ---- a received piece: the arrays live here until the upload has copied them */
struct piece_mem { void** blocks; int n, cap; };
static void* piece_alloc(struct piece_mem* pm, size_t bytes) {
  if (pm->n == pm->cap) {
    const int cap = pm->cap ? 2 * pm->cap : 256;
    void** b = (void**)realloc(pm->blocks, sizeof(void*) * (size_t)cap);
    if (!b) return NULL;
    pm->blocks = b; pm->cap = cap;
  }
  void* p = malloc(bytes ? bytes : 1);
  if (p) pm->blocks[pm->n++] = p;
  return p;
}
static void piece_free(struct piece_mem* pm) {
  for (int i = 0; i < pm->n; i++) free(pm->blocks[i]);
  free(pm->blocks);
  pm->blocks = NULL; pm->n = pm->cap = 0;
}
static const int64_t elem_size[3] = {4, 8, 8};
struct item_header { char name[96]; int32_t dtype; int32_t pad; int64_t count; };

/* rank 0 -> dst: every item of the archive; a header with count = -1 ends the sequence */
static int send_piece(airhip_archive ar, int dst, MPI_Comm comm) {
  int n = 0;
  if (airhip_archive_count(ar, &n) != AIRHIP_OK) return 1;
  for (int i = 0; i <= n; i++) {
    struct item_header hd;
    memset(&hd, 0, sizeof hd);
    const void* data = NULL;
    if (i < n) {
      const char* name; int dt; int64_t cnt;
      if (airhip_archive_item(ar, i, &name, &dt, &data, &cnt) != AIRHIP_OK) return 1;
      strncpy(hd.name, name, sizeof hd.name - 1); hd.dtype = dt; hd.count = cnt;
    } else hd.count = -1;
    if (MPI_Send(&hd, (int)sizeof hd, MPI_BYTE, dst, 78, comm) != MPI_SUCCESS) return 1;
    if (i < n && hd.count > 0 && send_big(data, hd.count * elem_size[hd.dtype], dst, 79, comm)) return 1;
  }
  return 0;
}
static int recv_piece(airhip_archive ar, struct piece_mem* pm, int src, MPI_Comm comm, int64_t* bytes) {
  for (;;) {
    struct item_header hd;
    if (MPI_Recv(&hd, (int)sizeof hd, MPI_BYTE, src, 78, comm, MPI_STATUS_IGNORE) != MPI_SUCCESS) return 1;
    if (hd.count == -1) return 0;
    if (hd.count < 0 || hd.dtype < 0 || hd.dtype > 2) return 1;      /* -2: rank 0 gave up */
    void* p = piece_alloc(pm, (size_t)(hd.count * elem_size[hd.dtype]));
    if (!p) return 1;
    if (hd.count > 0 && recv_big(p, hd.count * elem_size[hd.dtype], src, 79, comm)) return 1;
    hd.name[sizeof hd.name - 1] = 0;
    if (airhip_archive_add(ar, hd.name, hd.dtype, p, hd.count) != AIRHIP_OK) return 1;
    *bytes += hd.count * elem_size[hd.dtype];
  }
}

int airhip_mpi_pcair_setup(MPI_Comm user_comm, const airhip_csr* local, int nopts, const char* const* names,
                           const char* const* values, const int64_t* hint, int transport, int host_only,
                           airhip_mpi_pcair* out, double* stats5) {
  int rc = AIRHIP_OK;
  struct airhip_mpi_pcair_s* h = NULL;
  airhip_pcair pg = NULL;            /* rank 0: the global hierarchy */
  airhip_partition part = NULL;
  airhip_archive mine = NULL;
  struct piece_mem pm = {NULL, 0, 0};
  int64_t *g_rp = NULL, *g_hint = NULL, *counts = NULL;
  int32_t *g_ci = NULL, *owner = NULL;
  double* g_v = NULL;
  double t_setup = 0.0, t_ship = 0.0;
  int levels = 0;
  int next_to_serve = -1;            /* rank 0: ranks next_to_serve .. 1 still wait for their piece */
  int64_t piece_bytes = 0;
  if (!local || !out || local->m < 0 || (nopts > 0 && (!names || !values))) { snprintf(g_err, sizeof g_err, "bad argument"); return AIRHIP_ERR_ARG; }
  h = (struct airhip_mpi_pcair_s*)calloc(1, sizeof *h);
  if (!h) { snprintf(g_err, sizeof g_err, "out of memory"); return AIRHIP_ERR_INTERNAL; }
  h->comm = MPI_COMM_NULL;
  MPIC(MPI_Comm_dup(user_comm, &h->comm));
  MPIC(MPI_Comm_rank(h->comm, &h->rank));
  MPIC(MPI_Comm_size(h->comm, &h->size));
  h->m_local = local->m; h->n_global = local->n;
  {
    /* MPIAIJ ownership: rank r's rows follow rank r-1's */
    int64_t m = local->m, before = 0, total = 0;
    MPIC(MPI_Exscan(&m, &before, 1, MPI_INT64_T, MPI_SUM, h->comm));
    if (h->rank == 0) before = 0;
    MPIC(MPI_Allreduce(&m, &total, 1, MPI_INT64_T, MPI_SUM, h->comm));
    h->row_start = before;
    if (total != local->n) FAIL(AIRHIP_ERR_ARG, "the ranks own %lld rows in all but the operator has %lld columns", (long long)total, (long long)local->n);
  }
  const int64_t nnz_local = local->m > 0 ? local->rowptr[local->m] - local->rowptr[0] : 0;
  /* ---- gather the operator on rank 0 (row lengths, columns, values; rank order = global row order) */
  counts = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)h->size);
  if (!counts) FAIL(AIRHIP_ERR_INTERNAL, "out of memory");      /* (before any rank waits for another) */
  {
    int64_t mine2[2] = {local->m, nnz_local};
    MPIC(MPI_Gather(mine2, 2, MPI_INT64_T, counts, 2, MPI_INT64_T, 0, h->comm));
  }
  if (h->rank == 0) {
    int64_t nnz = 0;
    for (int r = 0; r < h->size; r++) nnz += counts[2 * r + 1];
    const int64_t n = local->n;
    g_rp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    g_ci = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    g_v = (double*)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    owner = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    if (hint) g_hint = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
  }
  int64_t* rel = NULL;               /* ranks > 0: row pointers relative to their first entry */
  if (h->rank != 0) rel = (int64_t*)malloc(sizeof(int64_t) * (size_t)(local->m ? local->m : 1));
  {
    /* every rank learns whether all of them have their buffers BEFORE the first point-to-point message: a rank that gave
     * up here used to leave the others blocked in their sends / receives */
    int ok = h->rank != 0 ? rel != NULL : (g_rp && g_ci && g_v && owner && (!hint || g_hint)), all_ok = 0;
    MPIC(MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, h->comm));
    if (!all_ok) { free(rel); FAIL(AIRHIP_ERR_INTERNAL, "out of memory gathering the operator on rank 0"); }
  }
  if (h->rank == 0) {
    const int64_t n = local->n;
    (void)n;
    int64_t row = 0, p = 0;
    g_rp[0] = 0;
    for (int r = 0; r < h->size; r++) {
      const int64_t mr = counts[2 * r], nr = counts[2 * r + 1];
      if (r == 0) {
        for (int64_t i = 0; i < mr; i++) g_rp[row + i + 1] = p + (local->rowptr[i + 1] - local->rowptr[0]);
        memcpy(g_ci + p, local->col + local->rowptr[0], sizeof(int32_t) * (size_t)nr);
        memcpy(g_v + p, local->val + local->rowptr[0], sizeof(double) * (size_t)nr);
        if (hint) memcpy(g_hint + row, hint, sizeof(int64_t) * (size_t)mr);
      } else {
        /* row pointers arrive relative to the sender's first entry */
        if (recv_big(g_rp + row + 1, 8 * mr, r, 70, h->comm)) FAIL(AIRHIP_ERR_INTERNAL, "receiving row pointers failed");
        for (int64_t i = 0; i < mr; i++) g_rp[row + i + 1] += p;
        if (recv_big(g_ci + p, 4 * nr, r, 71, h->comm) || recv_big(g_v + p, 8 * nr, r, 72, h->comm)) FAIL(AIRHIP_ERR_INTERNAL, "receiving operator entries failed");
        if (hint && recv_big(g_hint + row, 8 * mr, r, 73, h->comm)) FAIL(AIRHIP_ERR_INTERNAL, "receiving the ordering hint failed");
      }
      for (int64_t i = 0; i < mr; i++) owner[row + i] = r;
      row += mr; p += nr;
    }
  } else {
    for (int64_t i = 0; i < local->m; i++) rel[i] = local->rowptr[i + 1] - local->rowptr[0];
    int bad = send_big(rel, 8 * local->m, 0, 70, h->comm);
    free(rel);
    bad |= send_big(local->col + (local->m ? local->rowptr[0] : 0), 4 * nnz_local, 0, 71, h->comm);
    bad |= send_big(local->val + (local->m ? local->rowptr[0] : 0), 8 * nnz_local, 0, 72, h->comm);
    if (hint) bad |= send_big(hint, 8 * local->m, 0, 73, h->comm);
    if (bad) FAIL(AIRHIP_ERR_INTERNAL, "sending the local rows to rank 0 failed");
  }
  /* ---- rank 0: hierarchy, cut, ship */
  AIR(airhip_archive_create(&mine));
  if (h->rank == 0) {
    const double t0 = MPI_Wtime();
    next_to_serve = h->size - 1;
    airhip_csr G;
    G.m = local->n; G.n = local->n; G.rowptr = g_rp; G.col = g_ci; G.val = g_v;
    AIR(airhip_pcair_create(&pg));
    for (int i = 0; i < nopts; i++) AIR(airhip_pcair_set_option(pg, names[i], values[i]));
    AIR(airhip_pcair_set_option(pg, "device_upload", "0"));           /* the hierarchy is cut below, not uploaded whole */
    if (g_hint) AIR(airhip_pcair_set_ordering_hint(pg, g_hint, G.m));
    if (host_only) AIR(airhip_pcair_setup_host_only(pg, &G)); else AIR(airhip_pcair_setup(pg, &G));
    AIR(airhip_pcair_num_levels(pg, &levels));
    t_setup = MPI_Wtime() - t0;
    const double t1 = MPI_Wtime();
    AIR(airhip_partition_create(pg, &G, h->size, owner, &part));
    for (int r = h->size - 1; r >= 0; r--) {          /* rank 0's own piece last: it stays alive until the upload */
      airhip_archive ar = NULL;
      AIR(airhip_partition_extract(part, r, &ar));
      if (r == 0) {
        /* copy the views into this rank's archive memory so that the partition can be released */
        int n = 0;
        AIR(airhip_archive_count(ar, &n));
        for (int i = 0; i < n; i++) {
          const char* name; int dt; const void* data; int64_t cnt;
          AIR(airhip_archive_item(ar, i, &name, &dt, &data, &cnt));
          void* p = piece_alloc(&pm, (size_t)(cnt * elem_size[dt]));
          if (!p) FAIL(AIRHIP_ERR_INTERNAL, "out of memory keeping rank 0's piece");
          if (cnt) memcpy(p, data, (size_t)(cnt * elem_size[dt]));
          AIR(airhip_archive_add(mine, name, dt, p, cnt));
          piece_bytes += cnt * elem_size[dt];
        }
      } else if (send_piece(ar, r, h->comm)) {
        airhip_archive_destroy(&ar);
        FAIL(AIRHIP_ERR_INTERNAL, "sending the piece of rank %d failed", r);
      }
      airhip_archive_destroy(&ar);
      next_to_serve = r - 1;
    }
    airhip_partition_destroy(&part);
    airhip_pcair_destroy(&pg);
    free(g_rp); free(g_ci); free(g_v); free(owner); free(g_hint);
    g_rp = NULL; g_ci = NULL; g_v = NULL; owner = NULL; g_hint = NULL;
    t_ship = MPI_Wtime() - t1;
  } else {
    if (recv_piece(mine, &pm, 0, h->comm, &piece_bytes)) FAIL(AIRHIP_ERR_INTERNAL, "receiving this rank's piece failed");
  }
  MPIC(MPI_Bcast(&levels, 1, MPI_INT, 0, h->comm));
  /* ---- every rank: transport + upload */
  if (!host_only) {
    const double t2 = MPI_Wtime();
    if (transport == AIRHIP_MPI_TRANSPORT_RCCL) {
      char uid[128];
      memset(uid, 0, sizeof uid);
      if (h->rank == 0) AIR(airhip_comm_rccl_unique_id(uid));
      MPIC(MPI_Bcast(uid, 128, MPI_BYTE, 0, h->comm));
      AIR(airhip_comm_create_rccl(h->rank, h->size, uid, &h->tr));
    } else {
      AIR(airhip_comm_create_staged(h->rank, h->size, mpi_exchange, mpi_allreduce, h, &h->tr));
    }
    AIR(airhip_pcair_create(&h->pc));
    AIR(airhip_pcair_load_distributed(h->pc, mine, h->tr));
    t_ship += MPI_Wtime() - t2;
  }
  if (stats5) { stats5[0] = levels; stats5[1] = (double)local->n; stats5[2] = (double)piece_bytes; stats5[3] = t_setup; stats5[4] = t_ship; }
done:
  if (h && h->rank == 0 && rc != AIRHIP_OK) {
    /* the ranks that have not been served yet are waiting in recv_piece: tell them to give up */
    struct item_header hd;
    memset(&hd, 0, sizeof hd);
    hd.count = -2;
    for (int r = next_to_serve; r >= 1; r--) (void)MPI_Send(&hd, (int)sizeof hd, MPI_BYTE, r, 78, h->comm);
  }
  if (mine) airhip_archive_destroy(&mine);
  piece_free(&pm);
  if (part) airhip_partition_destroy(&part);
  if (pg) airhip_pcair_destroy(&pg);
  free(g_rp); free(g_ci); free(g_v); free(owner); free(g_hint); free(counts);
  {
    /* a failure on one rank must not leave the others waiting inside the next collective: agree on the outcome */
    int bad = rc != AIRHIP_OK, any = 0;
    if (h && h->comm != MPI_COMM_NULL && MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_MAX, h->comm) == MPI_SUCCESS && any && rc == AIRHIP_OK) {
      snprintf(g_err, sizeof g_err, "the distributed setup failed on another rank");
      rc = AIRHIP_ERR_INTERNAL;
    }
  }
  if (rc != AIRHIP_OK) { airhip_mpi_pcair_destroy(&h); return rc; }
  *out = h;
  return AIRHIP_OK;
}

int airhip_mpi_pcair_apply(airhip_mpi_pcair pc, const double* d_x, double* d_y) {
  if (!pc || !pc->pc) { snprintf(g_err, sizeof g_err, "no device hierarchy (host_only setup?)"); return AIRHIP_ERR_STATE; }
  int rc = airhip_pcair_apply(pc->pc, d_x, d_y);
  if (rc == AIRHIP_OK) rc = airhip_pcair_sync(pc->pc);
  if (rc != AIRHIP_OK) snprintf(g_err, sizeof g_err, "%s", airhip_last_error());
  return rc;
}

int airhip_mpi_ksp_solve(airhip_mpi_pcair pc, const airhip_ksp_options* o, const double* d_b, double* d_x, airhip_ksp_result* res) {
  if (!pc || !pc->pc) { snprintf(g_err, sizeof g_err, "no device hierarchy (host_only setup?)"); return AIRHIP_ERR_STATE; }
  const int rc = airhip_ksp_solve(o, NULL, pc->pc, NULL, d_b, d_x, res, NULL, 0, NULL);
  if (rc != AIRHIP_OK) snprintf(g_err, sizeof g_err, "%s", airhip_last_error());
  return rc;
}

int airhip_mpi_pcair_handle(airhip_mpi_pcair pc, airhip_pcair* inner, airhip_comm* comm) {
  if (!pc) { snprintf(g_err, sizeof g_err, "null argument"); return AIRHIP_ERR_ARG; }
  if (inner) *inner = pc->pc;
  if (comm) *comm = pc->tr;
  return AIRHIP_OK;
}

int airhip_mpi_pcair_traffic(airhip_mpi_pcair pc, int64_t* exchanges, int64_t* messages, int64_t* bytes, int64_t* allreduces) {
  if (!pc || !pc->tr) { snprintf(g_err, sizeof g_err, "no transport"); return AIRHIP_ERR_STATE; }
  int64_t ex = 0, ar = 0, by = 0, ms = 0;
  int rc = airhip_comm_stats(pc->tr, &ex, &ar);
  if (rc == AIRHIP_OK) rc = airhip_comm_traffic(pc->tr, &by, &ms);
  if (exchanges) *exchanges = ex;
  if (allreduces) *allreduces = ar;
  if (bytes) *bytes = by;
  if (messages) *messages = ms;
  return rc;
}

int airhip_mpi_pcair_destroy(airhip_mpi_pcair* pc) {
  if (!pc || !*pc) return AIRHIP_OK;
  struct airhip_mpi_pcair_s* h = *pc;
  if (h->pc) airhip_pcair_destroy(&h->pc);       /* before the communicator it uses */
  if (h->tr) airhip_comm_destroy(&h->tr);
  if (h->comm != MPI_COMM_NULL) MPI_Comm_free(&h->comm);
  free(h);
  *pc = NULL;
  return AIRHIP_OK;
}
