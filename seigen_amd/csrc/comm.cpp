// Native halo exchange: the facet-trace exchange between the blocks of a partitioned mesh, driven from inside the
// library over RCCL (one process per GPU, point-to-point between face neighbours over xGMI).
//
// In the reference the exchange is implicit in every assemble (seigen/elastic.py:364; halo depth set up in
// :404-436; PyOP2's ParLoopHaloEnd, tests/tiling/utils.py:144).  Here sg_step on a block with neighbours runs, for
// each of the six stages of every step and without returning to the host language:
//     FIRST (shell + half the interior, main stream)  ->  pack the OUTPUT's traces on all sides (one launch)
//     ->  ncclGroupStart / ncclSend + ncclRecv per face neighbour / ncclGroupEnd on the main stream
//     ->  SECOND (other half of the interior) on the second stream, beside the exchange
// and the next stage's FIRST follows the receives in stream order.  The Python exchanger (seigen_amd/parallel.py)
// stays for process groups that cannot move device memory (gloo: CPU tests, ranks sharing one GPU).
#include <rccl/rccl.h>

#include "handle.hpp"

struct sg_comm_state {
  ncclComm_t comm = nullptr;
  int rank = -1, nranks = 0;
  int peers[6] = {-1, -1, -1, -1, -1, -1};
  int nsides = 0, sides[6];
  void* send[2][6];   // [0: velocity-like fields, 1: stress-like fields][side]
  void* recv[2][6];
  size_t count[6];    // values per side (dim per facet node for either kind)
  sg_comm_stats_t stats;
  std::vector<hipEvent_t> wait_events;   // pairs (end of SECOND, end of the receives) still to be read
  std::vector<hipEvent_t> event_pool;
};

#define NCCLCHECK(h, expr)                                                                       \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      (h)->err = std::string(#expr) + ": " + ncclGetErrorString(_r);                             \
      return SG_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

static const int kStageOutput[6] = {SG_FIELD_UH, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_SH, SG_FIELD_UH, SG_FIELD_S};
static const int kStageInput[6] = {SG_FIELD_S, SG_FIELD_UH, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_SH, SG_FIELD_UH};

static int resolve_waits(sg_handle* h) {
  sg_comm_state* c = h->comm;
  for (size_t k = 0; k + 1 < c->wait_events.size(); k += 2) {
    HIPCHECK(h, hipEventSynchronize(c->wait_events[k]));
    HIPCHECK(h, hipEventSynchronize(c->wait_events[k + 1]));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, c->wait_events[k], c->wait_events[k + 1]));
    if (ms > 0) c->stats.exposed_wait_ms += ms;
    c->event_pool.push_back(c->wait_events[k]);
    c->event_pool.push_back(c->wait_events[k + 1]);
  }
  c->wait_events.clear();
  return SG_OK;
}

static int take_event(sg_handle* h, hipEvent_t* e) {
  sg_comm_state* c = h->comm;
  if (!c->event_pool.empty()) {
    *e = c->event_pool.back();
    c->event_pool.pop_back();
    return SG_OK;
  }
  HIPCHECK(h, hipEventCreate(e));
  return SG_OK;
}

// pack the traces of `field` on every side that has a neighbour and post the sends / receives (main stream)
static int exchange(sg_handle* h, int field, hipEvent_t* recv_done) {
  sg_comm_state* c = h->comm;
  const int kind = field_is_stress(field) ? 1 : 0;
  void* outs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < c->nsides; ++i) outs[c->sides[i]] = c->send[kind][c->sides[i]];
  int rc = sg_halo_pack_sides(h, field, outs);
  if (rc != SG_OK) return rc;
  const ncclDataType_t ty = h->f32 ? ncclFloat : ncclDouble;
  static const bool dry = std::getenv("SEIGEN_COMM_DRY") != nullptr;   // measurements only: everything but the transport
  if (dry) {
    c->stats.exchanges += 1;
    if (recv_done) {
      rc = take_event(h, recv_done);
      if (rc != SG_OK) return rc;
      HIPCHECK(h, hipEventRecord(*recv_done, h->stream));
    }
    return SG_OK;
  }
  NCCLCHECK(h, ncclGroupStart());
  for (int i = 0; i < c->nsides; ++i) {
    const int s = c->sides[i];
    NCCLCHECK(h, ncclSend(c->send[kind][s], c->count[s], ty, c->peers[s], c->comm, h->stream));
    NCCLCHECK(h, ncclRecv(c->recv[kind][s], c->count[s], ty, c->peers[s], c->comm, h->stream));
    c->stats.bytes_sent += (int64_t)(c->count[s] * (h->f32 ? sizeof(float) : sizeof(double)));
  }
  NCCLCHECK(h, ncclGroupEnd());
  c->stats.exchanges += 1;
  if (recv_done) {
    rc = take_event(h, recv_done);
    if (rc != SG_OK) return rc;
    HIPCHECK(h, hipEventRecord(*recv_done, h->stream));
  }
  return SG_OK;
}

// sg_step of a block with neighbours (stages.cpp): the pipelined schedule, `nsteps` whole steps
int comm_step(sg_handle* h, int64_t nsteps) {
  sg_comm_state* c = h->comm;
  if (nsteps <= 0) return SG_OK;
  HIPCHECK(h, hipEventRecord(h->ev0, h->stream));
  // the halo of the first stage's input: the caller may have changed the fields since the last call
  int rc = exchange(h, kStageInput[0], nullptr);
  if (rc != SG_OK) return rc;
  for (int64_t k = 0; k < nsteps; ++k) {
    for (int st = 0; st < 6; ++st) {
      rc = sg_run_stage(h, st, SG_REGION_FIRST);
      if (rc != SG_OK) return rc;
      hipEvent_t recv_done = nullptr, second_done = nullptr;
      rc = exchange(h, kStageOutput[st], h->timing ? &recv_done : nullptr);
      if (rc != SG_OK) return rc;
      rc = sg_run_stage(h, st, SG_REGION_SECOND);
      if (rc != SG_OK) return rc;
      if (h->timing) {
        // what the next stage waits for the traces BEYOND the end of the SECOND launch that ran beside them
        rc = take_event(h, &second_done);
        if (rc != SG_OK) return rc;
        HIPCHECK(h, hipEventRecord(second_done, h->overlap ? h->stream2 : h->stream));
        c->wait_events.push_back(second_done);
        c->wait_events.push_back(recv_done);
        if (c->wait_events.size() >= 8192) {
          rc = resolve_waits(h);
          if (rc != SG_OK) return rc;
        }
      }
    }
    rc = sg_end_step(h);
    if (rc != SG_OK) return rc;
  }
  if (int rc2 = join_second(h)) return rc2;
  HIPCHECK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHECK(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  return SG_OK;
}

void comm_release(sg_handle* h) {
  sg_comm_state* c = h->comm;
  if (!c) return;
  (void)sync_all(h);
  for (int k = 0; k < 2; ++k)
    for (int s = 0; s < 6; ++s) {
      if (c->send[k][s]) (void)hipFree(c->send[k][s]);
      if (c->recv[k][s]) (void)hipFree(c->recv[k][s]);
    }
  for (hipEvent_t e : c->wait_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  for (int f = 0; f < 4; ++f)
    for (int s = 0; s < 6; ++s) h->ghost[f][s] = nullptr;
  delete c;
  h->comm = nullptr;
}

extern "C" {

int sg_comm_get_unique_id(void* id, size_t nbytes) {
  if (!id || nbytes != SG_COMM_ID_BYTES) return SG_ERR_ARG;
  static_assert(SG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "seigen_hip.h and rccl.h disagree on the size of a unique id");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return SG_ERR_DEVICE;
  std::memcpy(id, u.internal, SG_COMM_ID_BYTES);
  return SG_OK;
}

int sg_comm_init(sg_handle* h, const void* id, size_t nbytes, int rank, int nranks, const int32_t* peers) {
  if (!h || !id || !peers || nbytes != SG_COMM_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks) return SG_ERR_ARG;
  if (h->comm) return fail(h, SG_ERR_STATE, "sg_comm_init: the handle already has a communicator");
  const int d = h->cfg.dim;
  for (int s = 0; s < 6; ++s) {
    const bool nbr = s < 2 * d && h->md.has_nbr[s];
    const int p = s < 2 * d ? peers[s] : -1;
    if (nbr != (p >= 0) || p >= nranks)
      return fail(h, SG_ERR_ARG, "sg_comm_init: peers[] must name a rank for exactly the sides of sg_config::nbr_mask");
  }
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  sg_comm_state* c = new sg_comm_state();
  std::memset(c->send, 0, sizeof(c->send));
  std::memset(c->recv, 0, sizeof(c->recv));
  std::memset(&c->stats, 0, sizeof(c->stats));
  h->comm = c;
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId u;
  std::memcpy(u.internal, id, SG_COMM_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    comm_release(h);
    return fail(h, SG_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  const size_t es = h->f32 ? sizeof(float) : sizeof(double);
  for (int s = 0; s < 2 * d; ++s) {
    c->peers[s] = peers[s];
    if (peers[s] < 0) continue;
    c->sides[c->nsides++] = s;
    size_t nb = 0;
    (void)sg_halo_bytes(h, SG_FIELD_U, s, &nb);
    c->count[s] = nb / es;
    for (int k = 0; k < 2; ++k) {
      if (hipMalloc(&c->send[k][s], nb) != hipSuccess || hipMalloc(&c->recv[k][s], nb) != hipSuccess) {
        comm_release(h);
        return fail(h, SG_ERR_NOMEM, "hipMalloc of a halo buffer failed");
      }
      (void)hipMemset(c->send[k][s], 0, nb);
      (void)hipMemset(c->recv[k][s], 0, nb);
    }
    // both fields of a kind read the same ghost buffer: it is consumed by the stage that follows its exchange
    // before the next exchange of that kind starts
    h->ghost[SG_FIELD_U][s] = h->ghost[SG_FIELD_UH][s] = (const double*)c->recv[0][s];
    h->ghost[SG_FIELD_S][s] = h->ghost[SG_FIELD_SH][s] = (const double*)c->recv[1][s];
  }
  HIPCHECK(h, hipDeviceSynchronize());
  return SG_OK;
}

int sg_comm_finalize(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  comm_release(h);
  return SG_OK;
}

int sg_comm_get_stats(sg_handle* h, sg_comm_stats_t* out, int reset) {
  if (!h || !out) return SG_ERR_ARG;
  if (!h->comm) return fail(h, SG_ERR_STATE, "no communicator (sg_comm_init)");
  int rc = resolve_waits(h);
  if (rc != SG_OK) return rc;
  *out = h->comm->stats;
  if (reset) std::memset(&h->comm->stats, 0, sizeof(h->comm->stats));
  return SG_OK;
}

// one exchange on its own (tests): pack `field` on every side with a neighbour, send, receive
int sg_comm_exchange(sg_handle* h, int field) {
  if (!h || field < 0 || field > 3) return SG_ERR_ARG;
  if (!h->comm) return fail(h, SG_ERR_STATE, "no communicator (sg_comm_init)");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return exchange(h, field, nullptr);
}

// DEVICE addresses of the send / receive buffer of a side (tests: what arrived); kind 0 = velocity-like, 1 = stress-like
int sg_comm_buffers(sg_handle* h, int kind, int side, void** send, void** recv, size_t* nbytes) {
  if (!h || kind < 0 || kind > 1 || side < 0 || side > 5) return SG_ERR_ARG;
  if (!h->comm) return fail(h, SG_ERR_STATE, "no communicator (sg_comm_init)");
  if (send) *send = h->comm->send[kind][side];
  if (recv) *recv = h->comm->recv[kind][side];
  if (nbytes) *nbytes = h->comm->count[side] * (h->f32 ? sizeof(float) : sizeof(double));
  return SG_OK;
}

}  // extern "C"
