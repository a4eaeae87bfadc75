/*
 * fake_rccl.c — TEST INFRASTRUCTURE: a stand-in for librccl.so that lets world > 1 run on ONE GPU (or on none).
 *
 * Real RCCL refuses two ranks on the same device, and the build container has no GPU at all, so the code around the
 * library's collectives (rsf_comm_init / rsf_comm_init_all, the bytes*world staging, the rank-major receive layout, the
 * destroy / re-init order) could otherwise only ever execute on the driver's 8-GPU node.  This file implements the nine
 * nccl* entry points the product binds (csrc/rsf_hip.hip, struct Rccl) with the semantics of the real ones as far as a
 * caller can observe them:
 *   - ncclCommInitRank blocks until all `nranks` ranks holding the same unique id have joined (one thread per rank);
 *   - ncclCommInitAll makes all ranks at once for a single thread;
 *   - a collective called OUTSIDE a group blocks until every rank of the communicator has called it (real RCCL enqueues
 *     and returns; waiting is a legal schedule of that), called INSIDE ncclGroupStart/End it is posted and the last rank's
 *     post performs it — a single thread can therefore post all ranks between GroupStart and GroupEnd, as with RCCL;
 *   - all ranks must agree on collective, count and type, else ncclInvalidArgument.
 * Data movement: memcpy for host pointers; with FAKE_RCCL_HIP=1 the pointers are device pointers and the copies go
 * through the process's own HIP runtime (hipStreamSynchronize on every rank's stream first, then hipMemcpy), looked up
 * with dlsym so that this file needs no ROCm header and no link-time dependency.  Selected through the product's
 * RSF_RCCL_LIB hook.  Types are laid out like <rccl/rccl.h>'s (ncclUniqueId: 128 bytes by value; handles: pointers).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef void *hipStream_t;

enum { MAX_RANKS = 64, KIND_ALLGATHER = 1, KIND_ALLREDUCE = 2 };

typedef struct {
  int kind;
  const void *send;
  void *recv;
  size_t count;
  hipStream_t stream;
} op_t;

typedef struct clique {
  ncclUniqueId id;
  int nranks, joined, left;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  op_t ops[MAX_RANKS];
  int posted_mask_count;
  char posted[MAX_RANKS];
  unsigned long generation;
  ncclResult_t last;
  struct clique *next;
} clique_t;

typedef struct fake_comm {
  clique_t *cl;
  int rank;
} *ncclComm_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static clique_t *g_cliques = NULL;
static unsigned long g_ids = 0;
static __thread int t_group = 0;

/* ---- data movement ---- */
typedef int (*hip_memcpy_fn)(void *, const void *, size_t, int);
typedef int (*hip_sync_fn)(hipStream_t);
static hip_memcpy_fn p_memcpy = NULL;
static hip_sync_fn p_sync = NULL;
static int g_hip = -1;

static int use_hip(void) {
  if (g_hip < 0) {
    const char *e = getenv("FAKE_RCCL_HIP");
    g_hip = (e && *e == '1') ? 1 : 0;
    if (g_hip) {
      p_memcpy = (hip_memcpy_fn)dlsym(RTLD_DEFAULT, "hipMemcpy");
      p_sync = (hip_sync_fn)dlsym(RTLD_DEFAULT, "hipStreamSynchronize");
      if (!p_memcpy || !p_sync) { fprintf(stderr, "fake_rccl: FAKE_RCCL_HIP=1 but the process holds no HIP runtime\n"); g_hip = 0; }
    }
  }
  return g_hip;
}

static int copy_bytes(void *dst, const void *src, size_t n) {
  if (dst == src || n == 0) return 0;
  if (use_hip()) return p_memcpy(dst, src, n, 4 /* hipMemcpyDefault */);
  memmove(dst, src, n);
  return 0;
}

static int sync_stream(hipStream_t s) { return use_hip() ? p_sync(s) : 0; }

/* every rank has posted: do the collective (called with cl->mu held) */
static ncclResult_t perform(clique_t *cl) {
  const int n = cl->nranks;
  const op_t *o = cl->ops;
  for (int r = 1; r < n; ++r)
    if (o[r].kind != o[0].kind || o[r].count != o[0].count) return ncclInvalidArgument;
  for (int r = 0; r < n; ++r)
    if (sync_stream(o[r].stream)) return ncclUnhandledCudaError;  /* everything the ranks enqueued before is done */
  const size_t bytes = o[0].count * sizeof(double);
  if (o[0].kind == KIND_ALLGATHER) {
    /* (an in-place call — send == recv + rank*count — leaves the rank's own block where it is: copy_bytes skips it) */
    for (int dst = 0; dst < n; ++dst)
      for (int src = 0; src < n; ++src)
        if (copy_bytes((char *)o[dst].recv + (size_t)src * bytes, o[src].send, bytes)) return ncclUnhandledCudaError;
  } else {
    double *acc = (double *)calloc(o[0].count ? o[0].count : 1, sizeof(double));
    double *tmp = (double *)malloc(bytes ? bytes : 8);
    if (!acc || !tmp) { free(acc); free(tmp); return ncclSystemError; }
    for (int r = 0; r < n; ++r) {  /* rank order: a deterministic sum */
      if (use_hip() ? p_memcpy(tmp, o[r].send, bytes, 4) : (memcpy(tmp, o[r].send, bytes), 0)) { free(acc); free(tmp); return ncclUnhandledCudaError; }
      for (size_t k = 0; k < o[0].count; ++k) acc[k] += tmp[k];
    }
    for (int r = 0; r < n; ++r)
      if (use_hip() ? p_memcpy(o[r].recv, acc, bytes, 4) : (memcpy(o[r].recv, acc, bytes), 0)) { free(acc); free(tmp); return ncclUnhandledCudaError; }
    free(acc);
    free(tmp);
  }
  return ncclSuccess;
}

static ncclResult_t post(ncclComm_t comm, op_t op) {
  if (!comm || !comm->cl) return ncclInvalidArgument;
  clique_t *cl = comm->cl;
  ncclResult_t res = ncclSuccess;
  pthread_mutex_lock(&cl->mu);
  if (cl->posted[comm->rank]) { pthread_mutex_unlock(&cl->mu); return ncclInvalidUsage; }  /* same rank twice in one round */
  cl->ops[comm->rank] = op;
  cl->posted[comm->rank] = 1;
  if (++cl->posted_mask_count == cl->nranks) {
    cl->last = perform(cl);
    memset(cl->posted, 0, sizeof cl->posted);
    cl->posted_mask_count = 0;
    ++cl->generation;
    res = cl->last;
    pthread_cond_broadcast(&cl->cv);
  } else if (t_group == 0) {  /* outside a group: wait for the other ranks (threads) */
    const unsigned long gen = cl->generation;
    while (cl->generation == gen) pthread_cond_wait(&cl->cv, &cl->mu);
    res = cl->last;
  }
  pthread_mutex_unlock(&cl->mu);
  return res;
}

/* ---- the nccl* entry points the product binds ---- */
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  pthread_mutex_lock(&g_mu);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-%lu-%p", ++g_ids, (void *)id);
  pthread_mutex_unlock(&g_mu);
  return ncclSuccess;
}

static clique_t *new_clique(const ncclUniqueId *id, int nranks) {
  clique_t *cl = (clique_t *)calloc(1, sizeof *cl);
  if (!cl) return NULL;
  if (id) cl->id = *id;
  cl->nranks = nranks;
  pthread_mutex_init(&cl->mu, NULL);
  pthread_cond_init(&cl->cv, NULL);
  cl->next = g_cliques;
  g_cliques = cl;
  return cl;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm_t c = (ncclComm_t)calloc(1, sizeof *c);
  if (!c) return ncclSystemError;
  pthread_mutex_lock(&g_mu);
  clique_t *cl = g_cliques;
  while (cl && (cl->joined >= cl->nranks || memcmp(&cl->id, &id, sizeof id) != 0)) cl = cl->next;  /* an open group with this id */
  if (!cl) cl = new_clique(&id, nranks);
  if (!cl || cl->nranks != nranks) { pthread_mutex_unlock(&g_mu); free(c); return cl ? ncclInvalidArgument : ncclSystemError; }
  c->cl = cl;
  c->rank = rank;
  ++cl->joined;
  pthread_cond_broadcast(&g_cv);
  while (cl->joined < cl->nranks) pthread_cond_wait(&g_cv, &g_mu);  /* collective: returns when every rank is in */
  pthread_mutex_unlock(&g_mu);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
  (void)devlist;  /* the real library refuses duplicate devices; this one exists to allow them */
  if (!comms || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
  pthread_mutex_lock(&g_mu);
  clique_t *cl = new_clique(NULL, ndev);
  if (cl) cl->joined = ndev;
  pthread_mutex_unlock(&g_mu);
  if (!cl) return ncclSystemError;
  for (int r = 0; r < ndev; ++r) {
    comms[r] = (ncclComm_t)calloc(1, sizeof **comms);
    if (!comms[r]) return ncclSystemError;
    comms[r]->cl = cl;
    comms[r]->rank = r;
  }
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  pthread_mutex_lock(&g_mu);
  clique_t *cl = comm->cl;
  if (cl && ++cl->left == cl->nranks) {  /* the last rank out frees the group */
    clique_t **pp = &g_cliques;
    while (*pp && *pp != cl) pp = &(*pp)->next;
    if (*pp) *pp = cl->next;
    pthread_mutex_destroy(&cl->mu);
    pthread_cond_destroy(&cl->cv);
    free(cl);
  }
  pthread_mutex_unlock(&g_mu);
  free(comm);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++t_group; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void) {
  if (t_group <= 0) return ncclInvalidUsage;
  --t_group;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  if (!sendbuff || !recvbuff || datatype != ncclFloat64) return ncclInvalidArgument;
  op_t op = {KIND_ALLGATHER, sendbuff, recvbuff, sendcount, stream};
  return post(comm, op);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t redop,
                           ncclComm_t comm, hipStream_t stream) {
  if (!sendbuff || !recvbuff || datatype != ncclFloat64 || redop != ncclSum) return ncclInvalidArgument;
  op_t op = {KIND_ALLREDUCE, sendbuff, recvbuff, count, stream};
  return post(comm, op);
}

const char *ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP copy or synchronise failed";
    case ncclSystemError: return "fake_rccl: out of memory";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (ranks disagree on collective/count, or a bad pointer/type)";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: internal error";
  }
}
