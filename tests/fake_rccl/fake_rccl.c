/* fake_rccl - a TEST DOUBLE of the nine RCCL entry points csrc/comm.cpp binds (rccl() there): point-to-point
 * transfers between ranks over POSIX shared memory with host staging, so that the native in-library halo exchange can
 * run with 2, 4 and 8 ranks that SHARE ONE GPU (real RCCL refuses two ranks on one device) - and, with
 * FAKE_RCCL_HOST=1, between plain host buffers on a machine without a GPU (the test of the double itself).
 *
 * TEST INFRASTRUCTURE ONLY: it lives under tests/, is built by the tests (tests/fake_rccl/build.py), is never linked
 * into or loaded by the product unless a test names it in SEIGEN_RCCL_LIB, and moves data far slower than RCCL.
 *
 * Semantics kept (what comm.cpp's exchange relies on; rccl.h "Point-to-point" section):
 *   - ncclSend / ncclRecv between one pair of ranks pair up in POSTING ORDER, per direction;
 *   - operations between ncclGroupStart / ncclGroupEnd are issued together at the outermost GroupEnd, so a rank may
 *     post sends and receives to several peers - and to itself - in one group without deadlock;
 *   - stream order: everything queued on `stream` before the call is complete before a byte is read, the received
 *     bytes are in place before anything queued after the call runs (the double blocks the HOST where RCCL would
 *     block the stream: stricter, never weaker);
 *   - ncclCommInitRank is collective over the nranks ranks that hold the same unique id.
 * Stricter than RCCL on purpose: a receive whose byte count differs from the matching send, a transfer that does not
 * arrive within FAKE_RCCL_TIMEOUT_S (default 60 s) and a rank number claimed twice are ERRORS (and poison the
 * communicator for every rank) instead of hangs or silent truncation.
 *
 * Fault injection for the rank-agreement tests (read at ncclCommInitRank):
 *   FAKE_RCCL_FAIL_INIT=<rank>     that rank's ncclCommInitRank fails (after the collective part, so the others return)
 *   FAKE_RCCL_CORRUPT_RECV=<rank>  every message that rank receives has its first byte inverted
 * FAKE_RCCL_LOG=<prefix>: ncclCommDestroy writes "<prefix>.rank<r>" with the counts of what this rank moved.
 *
 * Several ranks may live in one process (one communicator per thread): group state is thread-local.
 *
 * FAKE_RCCL_ASYNC=1 (GPU mode, ONE rank per process): the transfers are enqueued on `stream` like RCCL's and the call
 * returns at once - device-to-host copy + a host function that publishes the message; a host function that waits for
 * the peer's message + host-to-device copy + a host function that frees the slot - so the library's own stream
 * dependencies (the SECOND launch on its second stream beside the exchange, the next stage behind the receives) are
 * exercised as under RCCL instead of being serialised by a blocking call.  The segment is pinned for that
 * (FAKE_RCCL_SLOT_BYTES <= 1 MiB there).  A rank per THREAD would deadlock on the runtime's one callback thread.
 */
#define _GNU_SOURCE
#include <rccl/rccl.h> /* types and the prototypes the definitions below must match */

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FK_MAGIC "FKRCCL1"
#define FK_NSLOT 8        /* messages in flight per (source, destination) */
#define FK_MAXRANKS 64
#define FK_PAGE 4096

typedef struct {
  _Atomic int nranks;     /* 0 until the first rank arrives */
  _Atomic int arrived;
  _Atomic int left;
  _Atomic int poisoned;
  _Atomic uint64_t slot_bytes;
  _Atomic int rank_taken[FK_MAXRANKS];
} fk_header;

typedef struct {
  _Atomic uint64_t head;  /* messages published by the source */
  _Atomic uint64_t tail;  /* messages consumed by the destination */
  uint64_t bytes[FK_NSLOT];
} fk_chan;

struct ncclComm {         /* rccl.h leaves the struct incomplete: this is the double's */
  int rank, nranks;
  int async, pinned;
  uint64_t send_enq[FK_MAXRANKS], recv_enq[FK_MAXRANKS];   /* async mode: messages ENQUEUED per peer (the slots they will use) */
  size_t slot_bytes, chan_stride, map_bytes;
  unsigned char* base;
  int corrupt_recv;
  uint64_t sends, recvs, bytes_sent, bytes_recv, groups;
  char name[NCCL_UNIQUE_ID_BYTES];
};

typedef struct {
  int is_send;
  void* buf;
  size_t bytes;
  int peer;
  ncclComm_t comm;
  hipStream_t stream;
} fk_op;

static __thread int tl_depth = 0;
static __thread fk_op* tl_ops = NULL;
static __thread int tl_nops = 0, tl_cap = 0;

/* ---- the HIP runtime, bound at run time (the host mode needs none) ---------------------------------------- */
typedef int (*fk_memcpy_fn)(void*, const void*, size_t, int);
typedef int (*fk_streamsync_fn)(void*);
typedef int (*fk_memcpy_async_fn)(void*, const void*, size_t, int, void*);
typedef int (*fk_hostfunc_fn)(void*, void (*)(void*), void*);
typedef int (*fk_hostreg_fn)(void*, size_t, unsigned);
typedef int (*fk_hostunreg_fn)(void*);
static fk_memcpy_fn fk_hipMemcpy = NULL;
static fk_streamsync_fn fk_hipStreamSynchronize = NULL;
static fk_memcpy_async_fn fk_hipMemcpyAsync = NULL;
static fk_hostfunc_fn fk_hipLaunchHostFunc = NULL;
static fk_hostreg_fn fk_hipHostRegister = NULL;
static fk_hostunreg_fn fk_hipHostUnregister = NULL;
static int fk_host_mode = -1;

static int fk_bind_hip(void) {
  if (fk_host_mode < 0) {
    const char* e = getenv("FAKE_RCCL_HOST");
    fk_host_mode = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
  }
  if (fk_host_mode) return 0;
  if (fk_hipMemcpy && fk_hipStreamSynchronize) return 0;
  void* lib = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    fprintf(stderr, "fake_rccl: libamdhip64.so could not be loaded: %s\n", dlerror());
    return -1;
  }
  fk_hipMemcpy = (fk_memcpy_fn)dlsym(lib, "hipMemcpy");
  fk_hipStreamSynchronize = (fk_streamsync_fn)dlsym(lib, "hipStreamSynchronize");
  fk_hipMemcpyAsync = (fk_memcpy_async_fn)dlsym(lib, "hipMemcpyAsync");
  fk_hipLaunchHostFunc = (fk_hostfunc_fn)dlsym(lib, "hipLaunchHostFunc");
  fk_hipHostRegister = (fk_hostreg_fn)dlsym(lib, "hipHostRegister");
  fk_hipHostUnregister = (fk_hostunreg_fn)dlsym(lib, "hipHostUnregister");
  return (fk_hipMemcpy && fk_hipStreamSynchronize) ? 0 : -1;
}

static double fk_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double fk_timeout(void) {
  const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
  double t = e ? atof(e) : 60.0;
  return t > 0 ? t : 60.0;
}

static void fk_pause(int spins) {
  if (spins < 200) return;
  struct timespec ts = {0, spins < 2000 ? 20000 : 200000};
  nanosleep(&ts, NULL);
}

static size_t fk_type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

static fk_header* fk_hdr(ncclComm_t c) { return (fk_header*)c->base; }
static fk_chan* fk_channel(ncclComm_t c, int src, int dst) {
  return (fk_chan*)(c->base + FK_PAGE + ((size_t)src * (size_t)c->nranks + (size_t)dst) * c->chan_stride);
}
static unsigned char* fk_slot(ncclComm_t c, fk_chan* ch, uint64_t seq) {
  return (unsigned char*)ch + FK_PAGE + (size_t)(seq % FK_NSLOT) * c->slot_bytes;
}

static ncclResult_t fk_poison(ncclComm_t c, const char* what, ncclResult_t r) {
  atomic_store(&fk_hdr(c)->poisoned, 1);
  fprintf(stderr, "fake_rccl[rank %d of %d]: %s\n", c->rank, c->nranks, what);
  return r;
}

/* ---- the entry points ---------------------------------------------------------------------------------------- */
ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = NCCL_VERSION_CODE;   /* the version of the header the product was compiled against */
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "unhandled system error (fake_rccl: timeout, shared memory, or an injected fault)";
    case ncclInternalError: return "internal error (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "unknown result code (fake_rccl)";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static _Atomic unsigned counter = 0;
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "%s/fkrccl_%ld_%u_%lx", FK_MAGIC, (long)getpid(),
           atomic_fetch_add(&counter, 1), (unsigned long)ts.tv_nsec ^ ((unsigned long)ts.tv_sec << 20));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > FK_MAXRANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (strncmp(id.internal, FK_MAGIC, strlen(FK_MAGIC)) != 0 || id.internal[NCCL_UNIQUE_ID_BYTES - 1] != 0) {
    fprintf(stderr, "fake_rccl: the unique id was not made by fake_rccl's ncclGetUniqueId\n");
    return ncclInvalidArgument;
  }
  if (fk_bind_hip() != 0) return ncclSystemError;
  ncclComm_t c = (ncclComm_t)calloc(1, sizeof(struct ncclComm));
  if (!c) return ncclSystemError;
  c->rank = rank;
  c->nranks = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal + strlen(FK_MAGIC));
  const char* sb = getenv("FAKE_RCCL_SLOT_BYTES");
  c->slot_bytes = sb ? (size_t)strtoull(sb, NULL, 10) : ((size_t)4 << 20);
  c->slot_bytes = (c->slot_bytes + FK_PAGE - 1) / FK_PAGE * FK_PAGE;
  c->chan_stride = FK_PAGE + FK_NSLOT * c->slot_bytes;
  c->map_bytes = FK_PAGE + (size_t)nranks * (size_t)nranks * c->chan_stride;
  const char* cr = getenv("FAKE_RCCL_CORRUPT_RECV");
  c->corrupt_recv = (cr && *cr && atoi(cr) == rank) ? 1 : 0;
  /* every rank creates-or-opens the segment and sizes it: the pages come up zero-filled, and all-zero IS the
   * initial state of the header and of every channel, so no rank has to go first */
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    fprintf(stderr, "fake_rccl: shm_open/ftruncate(%s, %zu): %s\n", c->name, c->map_bytes, strerror(errno));
    if (fd >= 0) close(fd);
    free(c);
    return ncclSystemError;
  }
  c->base = (unsigned char*)mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) {
    fprintf(stderr, "fake_rccl: mmap: %s\n", strerror(errno));
    free(c);
    return ncclSystemError;
  }
  fk_header* h = fk_hdr(c);
  ncclResult_t res = ncclSuccess;
  int expect = 0;
  if (!atomic_compare_exchange_strong(&h->nranks, &expect, nranks) && expect != nranks)
    res = fk_poison(c, "ranks disagree on nranks", ncclInvalidArgument);
  uint64_t expect_sb = 0;
  if (!atomic_compare_exchange_strong(&h->slot_bytes, &expect_sb, (uint64_t)c->slot_bytes) && expect_sb != c->slot_bytes)
    res = fk_poison(c, "ranks disagree on FAKE_RCCL_SLOT_BYTES", ncclInvalidArgument);
  if (atomic_exchange(&h->rank_taken[rank], 1) != 0) res = fk_poison(c, "rank number claimed twice", ncclInvalidArgument);
  atomic_fetch_add(&h->arrived, 1);
  const double t0 = fk_now(), limit = fk_timeout();
  int spins = 0;
  while (res == ncclSuccess && atomic_load(&h->arrived) < nranks) {
    if (atomic_load(&h->poisoned)) res = ncclSystemError;
    else if (fk_now() - t0 > limit) res = fk_poison(c, "ncclCommInitRank: the other ranks did not arrive", ncclSystemError);
    fk_pause(++spins);
  }
  if (res == ncclSuccess && atomic_load(&h->poisoned)) res = ncclSystemError;
  /* the name is not needed any more once everybody has mapped the segment (or nobody will): the mappings keep it alive */
  if (rank == 0 || res != ncclSuccess) shm_unlink(c->name);
  const char* as = getenv("FAKE_RCCL_ASYNC");
  if (res == ncclSuccess && as && *as && strcmp(as, "0") != 0 && !fk_host_mode) {
    if (!fk_hipMemcpyAsync || !fk_hipLaunchHostFunc || !fk_hipHostRegister || !fk_hipHostUnregister || c->slot_bytes > ((size_t)1 << 20)) {
      fprintf(stderr, "fake_rccl: FAKE_RCCL_ASYNC needs hipLaunchHostFunc / hipHostRegister and FAKE_RCCL_SLOT_BYTES <= 1 MiB\n");
      res = ncclInvalidUsage;
    } else if (fk_hipHostRegister(c->base, c->map_bytes, 0) != 0) {
      fprintf(stderr, "fake_rccl: hipHostRegister of the segment failed\n");
      res = ncclSystemError;
    } else {
      c->async = c->pinned = 1;
    }
  }
  const char* fi = getenv("FAKE_RCCL_FAIL_INIT");
  if (res == ncclSuccess && fi && *fi && atoi(fi) == rank) {
    fprintf(stderr, "fake_rccl[rank %d]: injected ncclCommInitRank failure\n", rank);
    res = ncclSystemError;     /* the other ranks have returned success: a failure of one rank alone */
  }
  if (res != ncclSuccess) {
    if (c->pinned) fk_hipHostUnregister(c->base);
    munmap(c->base, c->map_bytes);
    free(c);
    return res;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  const char* log = getenv("FAKE_RCCL_LOG");
  if (log && *log) {
    char path[512];
    snprintf(path, sizeof(path), "%s.rank%d", log, c->rank);
    FILE* f = fopen(path, "a");
    if (f) {
      fprintf(f, "{\"rank\": %d, \"nranks\": %d, \"sends\": %llu, \"recvs\": %llu, \"bytes_sent\": %llu, \"bytes_recv\": %llu, "
                 "\"groups\": %llu, \"host_mode\": %d, \"async\": %d}\n",
              c->rank, c->nranks, (unsigned long long)c->sends, (unsigned long long)c->recvs,
              (unsigned long long)c->bytes_sent, (unsigned long long)c->bytes_recv, (unsigned long long)c->groups, fk_host_mode, c->async);
      fclose(f);
    }
  }
  atomic_fetch_add(&fk_hdr(c)->left, 1);
  if (c->pinned) fk_hipHostUnregister(c->base);
  munmap(c->base, c->map_bytes);
  free(c);
  return ncclSuccess;
}

static ncclResult_t fk_copy(void* dst, const void* src, size_t n, int kind /* 1 h2d, 2 d2h */) {
  if (n == 0) return ncclSuccess;
  if (fk_host_mode) {
    memcpy(dst, src, n);
    return ncclSuccess;
  }
  return fk_hipMemcpy(dst, src, n, kind) == 0 ? ncclSuccess : ncclUnhandledCudaError;
}

static ncclResult_t fk_do_send(fk_op* op) {
  ncclComm_t c = op->comm;
  fk_chan* ch = fk_channel(c, c->rank, op->peer);
  const uint64_t seq = atomic_load(&ch->head);
  const double t0 = fk_now(), limit = fk_timeout();
  int spins = 0;
  while (seq - atomic_load(&ch->tail) >= FK_NSLOT) {   /* more messages in flight than the double holds */
    if (atomic_load(&fk_hdr(c)->poisoned)) return ncclSystemError;
    if (fk_now() - t0 > limit) return fk_poison(c, "ncclSend: the destination does not receive", ncclSystemError);
    fk_pause(++spins);
  }
  ncclResult_t r = fk_copy(fk_slot(c, ch, seq), op->buf, op->bytes, 2);
  if (r != ncclSuccess) return fk_poison(c, "ncclSend: copy out of the send buffer failed", r);
  ch->bytes[seq % FK_NSLOT] = op->bytes;
  atomic_store_explicit(&ch->head, seq + 1, memory_order_release);
  c->sends += 1;
  c->bytes_sent += op->bytes;
  return ncclSuccess;
}

static ncclResult_t fk_do_recv(fk_op* op) {
  ncclComm_t c = op->comm;
  fk_chan* ch = fk_channel(c, op->peer, c->rank);
  const uint64_t seq = atomic_load(&ch->tail);
  const double t0 = fk_now(), limit = fk_timeout();
  int spins = 0;
  while (atomic_load_explicit(&ch->head, memory_order_acquire) <= seq) {
    if (atomic_load(&fk_hdr(c)->poisoned)) return ncclSystemError;
    if (fk_now() - t0 > limit) return fk_poison(c, "ncclRecv: nothing arrived from the source", ncclSystemError);
    fk_pause(++spins);
  }
  if (ch->bytes[seq % FK_NSLOT] != op->bytes) {
    char msg[160];
    snprintf(msg, sizeof(msg), "ncclRecv of %zu bytes from rank %d matched a send of %llu bytes", op->bytes, op->peer,
             (unsigned long long)ch->bytes[seq % FK_NSLOT]);
    return fk_poison(c, msg, ncclInvalidArgument);
  }
  unsigned char* slot = fk_slot(c, ch, seq);
  if (c->corrupt_recv && op->bytes) slot[0] = (unsigned char)~slot[0];
  ncclResult_t r = fk_copy(op->buf, slot, op->bytes, 1);
  if (r != ncclSuccess) return fk_poison(c, "ncclRecv: copy into the receive buffer failed", r);
  atomic_store_explicit(&ch->tail, seq + 1, memory_order_release);
  c->recvs += 1;
  c->bytes_recv += op->bytes;
  return ncclSuccess;
}

/* ---- async mode: the transfers as stream-ordered work ---------------------------------------------------------------- */
typedef struct {
  ncclComm_t comm;
  fk_chan* ch;
  uint64_t seq;
  size_t bytes;
  int peer;
} fk_cb;

static void fk_cb_publish(void* p) {   /* runs behind the device-to-host copy of the message */
  fk_cb* a = (fk_cb*)p;
  a->ch->bytes[a->seq % FK_NSLOT] = a->bytes;
  atomic_store_explicit(&a->ch->head, a->seq + 1, memory_order_release);
  free(a);
}

static void fk_cb_wait(void* p) {      /* holds the stream until the peer's message is there */
  fk_cb* a = (fk_cb*)p;
  ncclComm_t c = a->comm;
  const double t0 = fk_now(), limit = fk_timeout();
  int spins = 0;
  while (atomic_load_explicit(&a->ch->head, memory_order_acquire) <= a->seq) {
    if (atomic_load(&fk_hdr(c)->poisoned)) break;
    if (fk_now() - t0 > limit) {
      fk_poison(c, "ncclRecv (async): nothing arrived from the source", ncclSystemError);
      break;
    }
    fk_pause(++spins);
  }
  if (!atomic_load(&fk_hdr(c)->poisoned) && a->ch->bytes[a->seq % FK_NSLOT] != a->bytes)
    fk_poison(c, "ncclRecv (async): byte count differs from the matching send", ncclInvalidArgument);
  if (c->corrupt_recv && a->bytes) {
    unsigned char* slot = fk_slot(c, a->ch, a->seq);
    slot[0] = (unsigned char)~slot[0];
  }
  free(a);
}

static void fk_cb_consume(void* p) {   /* runs behind the host-to-device copy: the slot is free again */
  fk_cb* a = (fk_cb*)p;
  atomic_store_explicit(&a->ch->tail, a->seq + 1, memory_order_release);
  free(a);
}

static fk_cb* fk_cb_new(ncclComm_t c, fk_chan* ch, uint64_t seq, size_t bytes, int peer) {
  fk_cb* a = (fk_cb*)malloc(sizeof(fk_cb));
  if (a) {
    a->comm = c;
    a->ch = ch;
    a->seq = seq;
    a->bytes = bytes;
    a->peer = peer;
  }
  return a;
}

static ncclResult_t fk_enqueue_send(fk_op* op) {
  ncclComm_t c = op->comm;
  fk_chan* ch = fk_channel(c, c->rank, op->peer);
  const uint64_t seq = c->send_enq[op->peer]++;
  const double t0 = fk_now(), limit = fk_timeout();
  int spins = 0;
  while (seq - atomic_load_explicit(&ch->tail, memory_order_acquire) >= FK_NSLOT) {   /* back-pressure: the host may run far ahead */
    if (atomic_load(&fk_hdr(c)->poisoned)) return ncclSystemError;
    if (fk_now() - t0 > limit) return fk_poison(c, "ncclSend (async): the destination does not receive", ncclSystemError);
    fk_pause(++spins);
  }
  fk_cb* a = fk_cb_new(c, ch, seq, op->bytes, op->peer);
  if (!a) return ncclSystemError;
  if ((op->bytes && fk_hipMemcpyAsync(fk_slot(c, ch, seq), op->buf, op->bytes, 2, (void*)op->stream) != 0) ||
      fk_hipLaunchHostFunc((void*)op->stream, fk_cb_publish, a) != 0)
    return fk_poison(c, "ncclSend (async): enqueue failed", ncclUnhandledCudaError);
  c->sends += 1;
  c->bytes_sent += op->bytes;
  return ncclSuccess;
}

static ncclResult_t fk_enqueue_recv(fk_op* op) {
  ncclComm_t c = op->comm;
  fk_chan* ch = fk_channel(c, op->peer, c->rank);
  const uint64_t seq = c->recv_enq[op->peer]++;
  fk_cb* a = fk_cb_new(c, ch, seq, op->bytes, op->peer);
  fk_cb* b = fk_cb_new(c, ch, seq, op->bytes, op->peer);
  if (!a || !b) return ncclSystemError;
  if (fk_hipLaunchHostFunc((void*)op->stream, fk_cb_wait, a) != 0 ||
      (op->bytes && fk_hipMemcpyAsync(op->buf, fk_slot(c, ch, seq), op->bytes, 1, (void*)op->stream) != 0) ||
      fk_hipLaunchHostFunc((void*)op->stream, fk_cb_consume, b) != 0)
    return fk_poison(c, "ncclRecv (async): enqueue failed", ncclUnhandledCudaError);
  c->recvs += 1;
  c->bytes_recv += op->bytes;
  return ncclSuccess;
}

static ncclResult_t fk_flush(void) {
  ncclResult_t res = ncclSuccess;
  if (tl_nops > 0 && tl_ops[0].comm && tl_ops[0].comm->async) {
    /* every send of the group before any receive, all of it stream-ordered: nothing here waits for the device */
    for (int i = 0; i < tl_nops && res == ncclSuccess; ++i)
      if (tl_ops[i].is_send) res = fk_enqueue_send(&tl_ops[i]);
    for (int i = 0; i < tl_nops && res == ncclSuccess; ++i)
      if (!tl_ops[i].is_send) res = fk_enqueue_recv(&tl_ops[i]);
    tl_ops[0].comm->groups += 1;
    tl_nops = 0;
    return res;
  }
  /* stream order, part 1: everything queued before the group has finished before a byte is read or overwritten */
  if (!fk_host_mode)
    for (int i = 0; i < tl_nops && res == ncclSuccess; ++i) {
      int seen = 0;
      for (int j = 0; j < i; ++j) seen |= tl_ops[j].stream == tl_ops[i].stream;
      if (!seen && fk_hipStreamSynchronize((void*)tl_ops[i].stream) != 0) res = ncclUnhandledCudaError;
    }
  /* every send of the group before any receive: no rank waits for a peer before its own data is on the way */
  for (int i = 0; i < tl_nops && res == ncclSuccess; ++i)
    if (tl_ops[i].is_send) res = fk_do_send(&tl_ops[i]);
  for (int i = 0; i < tl_nops && res == ncclSuccess; ++i)
    if (!tl_ops[i].is_send) res = fk_do_recv(&tl_ops[i]);
  /* part 2: the copies above are synchronous, so what is queued after the call sees the received bytes */
  if (tl_nops > 0 && tl_ops[0].comm) tl_ops[0].comm->groups += 1;
  tl_nops = 0;
  return res;
}

static ncclResult_t fk_post(int is_send, void* buf, size_t count, ncclDataType_t ty, int peer, ncclComm_t c, hipStream_t stream) {
  if (!c || peer < 0 || peer >= c->nranks || (count && !buf)) return ncclInvalidArgument;
  const size_t es = fk_type_bytes(ty);
  if (es == 0) return ncclInvalidArgument;
  if (count * es > c->slot_bytes) {
    fprintf(stderr, "fake_rccl: a message of %zu bytes exceeds FAKE_RCCL_SLOT_BYTES = %zu\n", count * es, c->slot_bytes);
    return ncclInvalidArgument;
  }
  if (atomic_load(&fk_hdr(c)->poisoned)) return ncclSystemError;
  if (tl_nops == tl_cap) {
    int cap = tl_cap ? 2 * tl_cap : 32;
    fk_op* p = (fk_op*)realloc(tl_ops, (size_t)cap * sizeof(fk_op));
    if (!p) return ncclSystemError;
    tl_ops = p;
    tl_cap = cap;
  }
  fk_op op = {is_send, buf, count * es, peer, c, stream};
  tl_ops[tl_nops++] = op;
  return tl_depth > 0 ? ncclSuccess : fk_flush();
}

ncclResult_t ncclGroupStart(void) {
  tl_depth += 1;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
  if (tl_depth <= 0) return ncclInvalidUsage;
  tl_depth -= 1;
  return tl_depth == 0 ? fk_flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t ty, int peer, ncclComm_t c, hipStream_t stream) {
  return fk_post(1, (void*)buf, count, ty, peer, c, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t ty, int peer, ncclComm_t c, hipStream_t stream) {
  return fk_post(0, buf, count, ty, peer, c, stream);
}
