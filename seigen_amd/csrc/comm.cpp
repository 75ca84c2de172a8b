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
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound at run time (rccl() below)

#include "handle.hpp"

// RCCL is bound lazily, on the first sg_comm_* call that needs it: libseigen_hip.so loads - and every single-block and
// device-free entry point works - on a machine without RCCL, and inside a process that has an RCCL already (torch
// brings its own copy) the calls resolve to THAT copy instead of mixing two versions in one process.  The entry
// points come from the copy already in the process (RTLD_DEFAULT) or, failing that, from librccl.so of the system -
// or, when SEIGEN_RCCL_LIB names a file, from that file and nothing else (a site's own RCCL build; the transport
// double of tests/fake_rccl).  Whatever was bound must report the major version of the <rccl/rccl.h> this file was
// compiled against (enum values and argument layouts are only checked at compile time otherwise), and the file it came
// from is recorded (sg_comm_library) so that a run can say which copy moved its traces.
namespace {
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
  int version = 0;
  std::string path;   // the shared object ncclCommInitRank lives in (dladdr)
  std::string err;
};

const RcclApi& rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    void* lib = RTLD_DEFAULT;
    if (const char* named = std::getenv("SEIGEN_RCCL_LIB"); named && *named) {
      lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
      if (!lib) {
        const char* why = dlerror();
        a.err = std::string("SEIGEN_RCCL_LIB=") + named + " could not be loaded: " + (why ? why : "?");
        return a;
      }
    } else if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
      lib = nullptr;
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
      if (!lib) {
        a.err = "RCCL is not available: librccl.so could not be loaded (multi-GPU runs need it; single blocks do not)";
        return a;
      }
    }
    bool all = true;
    auto bind = [&](auto& fn, const char* sym) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, sym));
      if (!fn) {
        all = false;
        a.err = std::string("RCCL entry point missing: ") + sym;
      }
    };
    bind(a.GetUniqueId, "ncclGetUniqueId");
    bind(a.CommInitRank, "ncclCommInitRank");
    bind(a.CommDestroy, "ncclCommDestroy");
    bind(a.GetErrorString, "ncclGetErrorString");
    bind(a.GroupStart, "ncclGroupStart");
    bind(a.GroupEnd, "ncclGroupEnd");
    bind(a.Send, "ncclSend");
    bind(a.Recv, "ncclRecv");
    bind(a.GetVersion, "ncclGetVersion");
    if (!all) return a;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(a.CommInitRank), &info) && info.dli_fname) a.path = info.dli_fname;
    if (a.GetVersion(&a.version) != ncclSuccess) {
      a.err = "ncclGetVersion failed (" + a.path + ")";
      return a;
    }
    // RCCL encodes 2.27.7 as 22707 (NCCL_VERSION, rccl.h; releases before 2.9 used major * 1000)
    const int major = a.version >= 10000 ? a.version / 10000 : a.version / 1000;
    if (major != NCCL_MAJOR) {
      a.err = "the RCCL in this process (" + a.path + ", version " + std::to_string(a.version) + ") has major version " +
              std::to_string(major) + ", this library was compiled against " + std::to_string(NCCL_MAJOR) + "." +
              std::to_string(NCCL_MINOR);
      return a;
    }
    a.ok = true;
    return a;
  }();
  return api;
}
}  // namespace

struct sg_comm_state {
  ncclComm_t comm = nullptr;
  int rank = -1, nranks = 0;
  int peers[6] = {-1, -1, -1, -1, -1, -1};
  int nsides = 0, sides[6];
  void* send[2][6];   // [0: velocity-like fields, 1: stress-like fields][side]
  void* recv[2][6];
  size_t count[6];    // values per side (dim per facet node for either kind)
  sg_comm_stats_t stats;
  bool used = false;   // field traces have been exchanged: the ghost buffers hold data a stage may still read
  std::vector<hipEvent_t> wait_events;   // pairs (end of SECOND, end of the receives) still to be read
  std::vector<hipEvent_t> event_pool;
};

#define NCCLCHECK(h, expr)                                                                       \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      (h)->err = std::string(#expr) + ": " + rccl().GetErrorString(_r);                          \
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

// pack the traces of `field` on every side that has a neighbour and post the sends / receives (main stream);
// pack = false: send what the send buffers hold (sg_comm_selftest)
static int exchange(sg_handle* h, int field, hipEvent_t* recv_done, bool pack = true) {
  sg_comm_state* c = h->comm;
  const int kind = field_is_stress(field) ? 1 : 0;
  int rc = SG_OK;
  if (pack) {
    void* outs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < c->nsides; ++i) outs[c->sides[i]] = c->send[kind][c->sides[i]];
    rc = sg_halo_pack_sides(h, field, outs);
    if (rc != SG_OK) return rc;
  }
  if (pack) c->used = true;
  const ncclDataType_t ty = h->f32 ? ncclFloat : ncclDouble;
#ifdef SG_COMM_DRY
  // experiment builds only (make EXTRA=-DSG_COMM_DRY, tools/experiments/neighbour_overhead2.py): everything but the
  // transport - the results of a block with neighbours are WRONG; never part of the shipped library
  c->stats.exchanges += 1;
  if (recv_done) {
    rc = take_event(h, recv_done);
    if (rc != SG_OK) return rc;
    HIPCHECK(h, hipEventRecord(*recv_done, h->stream));
  }
  return SG_OK;
#endif
  const RcclApi& nc = rccl();
  NCCLCHECK(h, nc.GroupStart());
  // sends in the order of my sides ...
  for (int s = 0; s < 6; ++s) {
    if (c->peers[s] < 0) continue;
    NCCLCHECK(h, nc.Send(c->send[kind][s], c->count[s], ty, c->peers[s], c->comm, h->stream));
    c->stats.bytes_sent += (int64_t)(c->count[s] * (h->f32 ? sizeof(float) : sizeof(double)));
  }
  // ... receives in the order of the FACING sides: RCCL pairs the sends and receives between two ranks in posting
  // order, and what belongs on my side s is what the neighbour sent from ITS side s ^ 1.  With one face per pair of
  // ranks the order is immaterial; with two (a block that is its own neighbour across an axis, two blocks around a
  // wrapped axis) side s would otherwise receive the peer's side-s trace - the mirror image, silently.
  for (int t = 0; t < 6; ++t) {
    const int s = t ^ 1;
    if (c->peers[s] < 0) continue;
    NCCLCHECK(h, nc.Recv(c->recv[kind][s], c->count[s], ty, c->peers[s], c->comm, h->stream));
  }
  NCCLCHECK(h, nc.GroupEnd());
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
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  for (int f = 0; f < 4; ++f)
    for (int s = 0; s < 6; ++s) h->ghost[f][s] = nullptr;
  delete c;
  h->comm = nullptr;
}

extern "C" {

int sg_comm_get_unique_id(void* id, size_t nbytes) {
  if (!id || nbytes != SG_COMM_ID_BYTES) return SG_ERR_ARG;
  static_assert(SG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "seigen_hip.h and rccl.h disagree on the size of a unique id");
  if (!rccl().ok) {
    g_create_err = rccl().err;
    return SG_ERR_STATE;
  }
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return SG_ERR_DEVICE;
  std::memcpy(id, u.internal, SG_COMM_ID_BYTES);
  return SG_OK;
}

int sg_comm_version(int* version) {
  if (!version) return SG_ERR_ARG;
  if (!rccl().ok) {
    g_create_err = rccl().err;
    return SG_ERR_STATE;
  }
  *version = rccl().version;
  return SG_OK;
}

// which RCCL moves the traces: the file the bound entry points come from (torch's bundled copy, the system's librccl.so,
// the file SEIGEN_RCCL_LIB names) - PyOP2 prints its MPI the same way in the reference's logs
int sg_comm_library(char* buf, size_t n) {
  if (!buf || n == 0) return SG_ERR_ARG;
  if (!rccl().ok) {
    g_create_err = rccl().err;
    return SG_ERR_STATE;
  }
  std::snprintf(buf, n, "%s", rccl().path.c_str());
  return SG_OK;
}

// everything sg_comm_init can refuse WITHOUT talking to another rank: the host calls it on every rank and agrees on
// the outcome before the collective ncclCommInitRank (a rank that failed here alone would leave the others waiting)
int sg_comm_check(sg_handle* h, int rank, int nranks, const int32_t* peers) {
  if (!h || !peers || nranks < 1 || rank < 0 || rank >= nranks) return SG_ERR_ARG;
  if (h->comm) return fail(h, SG_ERR_STATE, "sg_comm_init: the handle already has a communicator");
  if (!rccl().ok) return fail(h, SG_ERR_STATE, rccl().err);
  const int d = h->cfg.dim;
  for (int s = 0; s < 6; ++s) {
    const bool nbr = s < 2 * d && h->md.has_nbr[s];
    const int p = s < 2 * d ? peers[s] : -1;
    if (nbr != (p >= 0) || p >= nranks)
      return fail(h, SG_ERR_ARG, "sg_comm_init: peers[] must name a rank for exactly the sides of sg_config::nbr_mask");
  }
  // Two sides may lead to the same rank only as the two ends of ONE axis (a wrapped axis one or two blocks wide): the
  // exchange pairs such sends and receives by facing side (exchange()).  Anything else - one rank behind two different
  // axes - has no layout of blocks behind it.
  for (int s = 0; s < 2 * d; ++s)
    for (int t = s + 1; t < 2 * d; ++t)
      if (peers[s] >= 0 && peers[s] == peers[t] && (s >> 1) != (t >> 1))
        return fail(h, SG_ERR_ARG, "sg_comm_init: one rank named as the neighbour across two different axes");
  return SG_OK;
}

int sg_comm_init(sg_handle* h, const void* id, size_t nbytes, int rank, int nranks, const int32_t* peers) {
  if (!h || !id || !peers || nbytes != SG_COMM_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks) return SG_ERR_ARG;
  if (int rc = sg_comm_check(h, rank, nranks, peers)) return rc;
  const int d = h->cfg.dim;
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
  ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    comm_release(h);
    return fail(h, SG_ERR_DEVICE, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
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

// Does every side receive what the facing side of its neighbour sent?  Every rank fills its send buffers with a pattern
// that names (rank, side, kind, position), all exchange WITHOUT packing, and every rank checks what arrived on side s
// against the pattern of (peers[s], s ^ 1) - known locally, so no second transport is needed as a reference.  A crossed,
// mis-ordered or mis-addressed exchange shows up as mismatches (collective: every rank of the communicator calls it).
static double selftest_value(int rank, int side, int kind, size_t i, bool f32) {
  const long tag = ((long)rank * 8 + side) * 2 + kind;
  return f32 ? (double)((tag % 4096) * 4096 + (long)(i % 4096)) : (double)(tag * 4294967296.0 + (double)(i % 4294967296ull));
}

int sg_comm_selftest(sg_handle* h, int64_t* mismatches) {
  if (!h || !mismatches) return SG_ERR_ARG;
  if (!h->comm) return fail(h, SG_ERR_STATE, "no communicator (sg_comm_init)");
  sg_comm_state* c = h->comm;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (int rc = join_second(h)) return rc;
  const size_t es = h->f32 ? sizeof(float) : sizeof(double);
  *mismatches = 0;
  // the test writes patterns into the send AND ghost buffers and leaves them zeroed: once traces have been exchanged a
  // stage may still read them, and zero traces would be a silently wrong run
  if (c->used)
    return fail(h, SG_ERR_STATE, "sg_comm_selftest overwrites the ghost buffers: call it right after sg_comm_init, before any exchange or step");
#ifdef SG_COMM_DRY
  return SG_OK;   // experiment builds without the transport: nothing to test
#endif
  const sg_comm_stats_t keep = c->stats;      // not part of the run's statistics
  std::vector<double> hd;
  std::vector<float> hf;
  for (int kind = 0; kind < 2; ++kind) {
    for (int s = 0; s < 6; ++s) {
      if (c->peers[s] < 0) continue;
      const size_t n = c->count[s];
      hd.resize(n);
      hf.resize(n);
      for (size_t i = 0; i < n; ++i) {
        hd[i] = selftest_value(c->rank, s, kind, i, h->f32 != 0);
        hf[i] = (float)hd[i];
      }
      HIPCHECK(h, hipMemcpy(c->send[kind][s], h->f32 ? (const void*)hf.data() : (const void*)hd.data(), n * es, hipMemcpyHostToDevice));
      HIPCHECK(h, hipMemset(c->recv[kind][s], 0xff, n * es));
    }
    int rc = exchange(h, kind ? SG_FIELD_S : SG_FIELD_U, nullptr, false);
    if (rc != SG_OK) return rc;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int s = 0; s < 6; ++s) {
      if (c->peers[s] < 0) continue;
      const size_t n = c->count[s];
      hd.resize(n);
      hf.resize(n);
      HIPCHECK(h, hipMemcpy(h->f32 ? (void*)hf.data() : (void*)hd.data(), c->recv[kind][s], n * es, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) {
        const double want = selftest_value(c->peers[s], s ^ 1, kind, i, h->f32 != 0);
        const double got = h->f32 ? (double)hf[i] : hd[i];
        if (!(got == want)) *mismatches += 1;
      }
      HIPCHECK(h, hipMemset(c->send[kind][s], 0, n * es));
      HIPCHECK(h, hipMemset(c->recv[kind][s], 0, n * es));
    }
  }
  c->stats = keep;
  return SG_OK;
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
