// C-ABI of libseigen_hip.so (see include/seigen_hip.h).  Host-side driver of the
// HIP stage kernels: owns device memory, the fused six-launch LF4 step
// (seigen/elastic.py:283-313) and the facet-trace halo buffers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/seigen_hip.h"
#include "kernels.hpp"
#include "mesh_tables.hpp"
#include "mfma_tables.hpp"
#include "refelem.hpp"

using namespace sg;

struct sg_handle {
  sg_config cfg;
  RefElem re;
  MeshDev md;
  MeshDev* md_dev = nullptr;
  double* Dt = nullptr;
  double* Lt = nullptr;
  double* field[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t field_len[4] = {0, 0, 0, 0};    // doubles, host layout (ncells * nd * comps)
  size_t field_alloc[4] = {0, 0, 0, 0};  // doubles allocated on the device (layout padding included)
  bool use_mfma = false;
  bool use_lane = false;
  bool use_tile = false;    // 2-D MFMA tile kernels (kernels_tile2d.hip), gw = 16
  int f32 = 0;              // sg_config.dtype = 1: fields, halo buffers, operator tiles and arithmetic are float (MFMA path)
  bool sym = false;         // MFMA path: all stress fields symmetric -> kernels touch only the i <= j lines
  int* sym_flag = nullptr;  // device word set by an upload that is not symmetric
  // active (cell group, class) items of each region of a split stage (MFMA / lane paths), by sg_region
  int32_t* region_items[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int32_t region_nitems[5] = {-1, -1, -1, -1, -1};  // -1: not built yet
  double* fragF = nullptr;  // MFMA operator fragment tables (device)
  double* fragG = nullptr;
  double* fragL = nullptr;
  double* staging = nullptr;  // host-layout staging buffer for layout conversion
  // large transfers: two pinned host slots + two device slots, so that the DMA of one chunk, the
  // layout kernel of the next and the host-side copy of the previous one overlap
  double* pin[2] = {nullptr, nullptr};
  double* dstage[2] = {nullptr, nullptr};
  hipEvent_t xfer_ev[2] = {nullptr, nullptr};
  unsigned long long* dbg = nullptr;  // SEIGEN_HIP_STAMPS=1 (diagnostic builds): [kind][8] cycle sums
  size_t staging_len = 0;
  int64_t ncells = 0;
  int ncls = 0;
  // parameters
  bool params_set = false;
  double rho = 1.0, dt = 0.0, lam0 = 0.0, mu0 = 0.0;
  int per_cell = 0;
  double* lam_d = nullptr;
  double* mu_d = nullptr;
  double* rho2_d = nullptr;  // per-cell density factors [cell][2] (kernels.hpp), or null
  int rho_physical = 0;      // scalar density: 0 = rho*u0 + ..., 1 = u0 + (...)/rho
  // sponge
  int32_t* sponge_slot = nullptr;
  double* sponge_B = nullptr;
  // source
  int64_t src_nnz = 0;
  int64_t src_nfirst = 0;  // source nodes are stored with those in cells of SG_REGION_FIRST first
  int64_t* src_nodes = nullptr;
  double* src_values = nullptr;  // [nsteps][nnz][dim*dim]
  int64_t src_nsteps = 0;
  int64_t src_step = 0;
  bool src_static = false;  // one time slice that holds at every step
  // 2-D tile path: the source is added inside the G stage kernels (StageArgs::src_slot / src_idx)
  bool src_fused = false;
  int32_t* src_slot_d = nullptr;
  int32_t* src_idx_d = nullptr;
  // halo
  const double* ghost[4][6];
  // execution
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Split stages (blocks with neighbours): SG_REGION_SECOND of a stage depends on the stage before it, not on the
  // FIRST launch of its own stage, so it may run on a second (lower-priority) stream and fill the slots that FIRST's
  // persistent blocks free as they drain (default; SEIGEN_HIP_OVERLAP=0: one stream).  ev_stage: everything before this stage's FIRST;
  // ev_second: the SECOND launch, which every later piece of work on `stream` waits for.
  bool overlap = false;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_stage = nullptr, ev_second = nullptr;
  bool second_pending = false;
  int grid_blocks = 0;  // persistent grid of the MFMA stage kernels: while an exchange is in flight ...
  int grid_full = 0;    // ... and otherwise (every block slot of the device)
  T2Const t2c;          // 2-D tile kernels: kernarg copy of the mesh tables
  int tile_grid = 0;    // 2-D tile kernels: cap of the grid in blocks of four waves (SEIGEN_HIP_TILE_GRID)
  // small blocks are launch-bound (config 1: six 5-us launches per step): sg_step replays captured
  // hipGraphs of one and of eight steps there; any setter that changes kernel arguments bumps the epoch
  bool graph_ok = false;
  uint64_t epoch = 0, graph_epoch = ~0ull;
  hipGraphExec_t graph1 = nullptr, graph8 = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;   // per-launch event pairs, resolved lazily (no sync in the hot loop)
  std::vector<int> ev_stage_ids;         // stage of pair k = events 2k, 2k+1
  sg_counters_t counters;
  std::string err;
};

// smallest 2-D block (cells) that takes the MFMA tile kernels instead of the generic kernel: they win at every
// size measured, 40 x 40 squares included (tools/path_sweep2d.py, profiles/r02/path_sweep2d_tile_v2.txt)
static constexpr int64_t SG_TILE2D_MIN_CELLS = 0;

static std::string g_create_err;
static_assert(SG_MAX_BOXES == SG_MAX_REGION_BOXES, "kernels.hpp and seigen_hip.h disagree on the box limit");

#define HIPCHECK(h, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                              \
      return SG_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

static int fail(sg_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

static bool field_is_stress(int f) { return f == SG_FIELD_S || f == SG_FIELD_SH; }

// work queued on `stream` from here on comes after the SECOND launch that may still run on stream2
static int join_second(sg_handle* h) {
  if (h->second_pending) {
    HIPCHECK(h, hipStreamWaitEvent(h->stream, h->ev_second, 0));
    h->second_pending = false;
  }
  return SG_OK;
}

// the host waits for everything the handle has queued (both streams)
static hipError_t sync_all(sg_handle* h) {
  if (h->second_pending) {
    hipError_t e = hipStreamWaitEvent(h->stream, h->ev_second, 0);
    if (e != hipSuccess) return e;
    h->second_pending = false;
  }
  return hipStreamSynchronize(h->stream);
}

extern "C" {

const char* sg_last_error(const sg_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

void sg_destroy(sg_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)sync_all(h);
  for (int f = 0; f < 4; ++f)
    if (h->field[f]) (void)hipFree(h->field[f]);
  if (h->md_dev) (void)hipFree(h->md_dev);
  if (h->Dt) (void)hipFree(h->Dt);
  if (h->Lt) (void)hipFree(h->Lt);
  if (h->fragF) (void)hipFree(h->fragF);
  if (h->fragG) (void)hipFree(h->fragG);
  if (h->fragL) (void)hipFree(h->fragL);
  if (h->staging) (void)hipFree(h->staging);
  for (int i = 0; i < 2; ++i) {
    if (h->pin[i]) (void)hipHostFree(h->pin[i]);
    if (h->dstage[i]) (void)hipFree(h->dstage[i]);
    if (h->xfer_ev[i]) (void)hipEventDestroy(h->xfer_ev[i]);
  }
  if (h->sym_flag) (void)hipFree(h->sym_flag);
  if (h->graph1) (void)hipGraphExecDestroy(h->graph1);
  if (h->graph8) (void)hipGraphExecDestroy(h->graph8);
  for (int r = 0; r < 5; ++r)
    if (h->region_items[r]) (void)hipFree(h->region_items[r]);
  if (h->dbg) {
    unsigned long long v[32];
    if (hipMemcpy(v, h->dbg, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess)
      for (int k = 0; k < 4; ++k)
        std::fprintf(stderr, "[seigen_hip stamps] %s mode %d: items %llu  cycles/item: setup %.0f volume %.0f lifts %.0f epilogue %.0f\n",
                     k < 2 ? "F" : "G", k & 1, v[8 * k + 4], v[8 * k + 4] ? (double)v[8 * k + 0] / v[8 * k + 4] : 0.0,
                     v[8 * k + 4] ? (double)v[8 * k + 1] / v[8 * k + 4] : 0.0, v[8 * k + 4] ? (double)v[8 * k + 2] / v[8 * k + 4] : 0.0,
                     v[8 * k + 4] ? (double)v[8 * k + 3] / v[8 * k + 4] : 0.0);
    (void)hipFree(h->dbg);
  }
  if (h->lam_d) (void)hipFree(h->lam_d);
  if (h->mu_d) (void)hipFree(h->mu_d);
  if (h->rho2_d) (void)hipFree(h->rho2_d);
  if (h->sponge_slot) (void)hipFree(h->sponge_slot);
  if (h->sponge_B) (void)hipFree(h->sponge_B);
  if (h->src_nodes) (void)hipFree(h->src_nodes);
  if (h->src_values) (void)hipFree(h->src_values);
  if (h->src_slot_d) (void)hipFree(h->src_slot_d);
  if (h->src_idx_d) (void)hipFree(h->src_idx_d);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->stream2) {
    (void)hipStreamSynchronize(h->stream2);
    (void)hipStreamDestroy(h->stream2);
  }
  if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
  if (h->ev_second) (void)hipEventDestroy(h->ev_second);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int create_impl(const sg_config* cfg, sg_handle* h) {
  if (cfg->dim < 1 || cfg->dim > 3) return fail(h, SG_ERR_ARG, "dim must be 1, 2 or 3");
  if (cfg->degree < 1 || cfg->degree > 4) return fail(h, SG_ERR_ARG, "degree must be 1..4");
  for (int a = 0; a < cfg->dim; ++a) {
    if (cfg->n[a] < 1) return fail(h, SG_ERR_ARG, "n[axis] must be >= 1");
    if (!(cfg->h[a] > 0.0)) return fail(h, SG_ERR_ARG, "h[axis] must be > 0");
  }
  h->cfg = *cfg;
  for (int a = cfg->dim; a < 3; ++a) {
    h->cfg.n[a] = 1;
    h->cfg.h[a] = 1.0;
    h->cfg.origin[a] = 0.0;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(h, SG_ERR_DEVICE, "no HIP device available (libseigen_hip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(h, SG_ERR_ARG, "device ordinal out of range");
  HIPCHECK(h, hipSetDevice(cfg->device));

  try {
    h->re = make_refelem(cfg->dim, cfg->degree);
    std::memset(&h->md, 0, sizeof(MeshDev));
    h->md.nd = h->re.nd;
    h->md.nf = h->re.nf;
    build_mesh_tables(cfg->dim, cfg->degree, cfg->diagonal, h->cfg.h, h->re.fnode.data(), h->re.lattice.data(), h->md);
  } catch (const std::exception& e) {
    return fail(h, SG_ERR_ARG, e.what());
  }
  for (int a = 0; a < 3; ++a) h->md.n[a] = h->cfg.n[a];
  for (int s = 0; s < 6; ++s) h->md.has_nbr[s] = (s < 2 * cfg->dim) ? ((cfg->nbr_mask >> s) & 1) : 0;
  h->ncls = h->md.ncls;
  h->ncells = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2] * h->ncls;
  // kernel path: MFMA kernels (interleaved layout) where they exist, unless SEIGEN_HIP_PATH=generic
  const char* path_env = std::getenv("SEIGEN_HIP_PATH");
  const bool force_generic = path_env && std::strcmp(path_env, "generic") == 0;
  // 3-D: the MFMA kernels at every degree (degrees 1 and 2 use 4x4x4 tiles only); measured with
  // tools/path_sweep.py they beat the lane and generic kernels everywhere except degree 1 on blocks
  // under 65536 cells (SEIGEN_HIP_PATH=mfma forces them)
  const bool force_mfma = path_env && std::strcmp(path_env, "mfma") == 0;
  const int64_t ncells_all = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2] * h->ncls;
  h->use_mfma = mfma_supported(cfg->dim, cfg->degree) && !force_generic &&
                !(path_env && std::strcmp(path_env, "lane") == 0) &&
                (cfg->degree >= 2 || ncells_all >= 65536 || force_mfma);
  // lane-per-cell kernels need enough 64-cell groups to fill the chip; below that the
  // thread-per-node generic kernel has more parallelism (SEIGEN_HIP_PATH=lane forces them)
  const bool force_lane = path_env && std::strcmp(path_env, "lane") == 0;
  const int64_t ncube_all = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2];
  h->use_lane = !h->use_mfma && lane_supported(cfg->dim, cfg->degree) && !force_generic &&
                (force_lane || ncube_all * h->ncls >= (cfg->degree == 1 ? 196608 : 120000));  // crossovers measured
                                                                     // (tools/path_sweep.py, profiles/r02/small_2d_configs_negative_results.txt)
  // 2-D: the MFMA tile kernels (16 cells per wave, operators in registers) from SG_TILE2D_MIN_CELLS cells up
  // (measured crossover against the generic kernel, tools/path_sweep.py); SEIGEN_HIP_PATH=tile forces them
  const bool force_tile = path_env && std::strcmp(path_env, "tile") == 0;
  h->use_tile = tile2d_supported(cfg->dim, cfg->degree) && !force_generic && !force_lane &&
                (force_tile || ncells_all >= SG_TILE2D_MIN_CELLS);
  if (h->use_tile) h->use_lane = false;
  if (cfg->dtype != 0 && cfg->dtype != 1) return fail(h, SG_ERR_ARG, "dtype must be 0 (f64) or 1 (f32)");
  h->f32 = cfg->dtype;
  if (h->f32 && !h->use_mfma)
    return fail(h, SG_ERR_ARG, "dtype f32 is implemented on the MFMA path (3-D blocks; degree 1 from 65536 cells)");
  h->md.gw = (h->use_mfma || h->use_tile) ? 16 : (h->use_lane ? 64 : 1);
  h->md.ncube = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2];
  h->md.ncube_pad = (h->md.ncube + h->md.gw - 1) / h->md.gw * h->md.gw;
  if (h->use_tile) h->t2c = tile2d_const(h->md);
  for (int f = 0; f < 4; ++f)
    for (int s = 0; s < 6; ++s) h->ghost[f][s] = nullptr;
  std::memset(&h->counters, 0, sizeof(h->counters));

  const int d = cfg->dim, nd = h->re.nd, nf = h->re.nf, nfaces = h->re.nfaces;
  // transposed operators: Dt[r][b][a], Lt[f][b'][a]
  std::vector<double> Dt((size_t)d * nd * nd), Lt((size_t)nfaces * nf * nd);
  for (int r = 0; r < d; ++r)
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b) Dt[((size_t)r * nd + b) * nd + a] = h->re.D[((size_t)r * nd + a) * nd + b];
  for (int f = 0; f < nfaces; ++f)
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nf; ++b) Lt[((size_t)f * nf + b) * nd + a] = h->re.L[((size_t)f * nd + a) * nf + b];
  if (h->use_lane) {
    // lane path: E_r = D_r - (L_0 R_0 - L_{r+1} R_{r+1}) / (2 (d-1)!) row-major (own-trace half of the
    // central flux folded into the volume operator, see mfma_tables.cpp), and L_f row-major
    double fact = 1.0;
    for (int i = 2; i <= d - 1; ++i) fact *= i;
    const double cfold = 1.0 / (2.0 * fact);
    for (int r = 0; r < d; ++r)
      for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nd; ++b) {
          double v = h->re.D[((size_t)r * nd + a) * nd + b];
          for (int bf = 0; bf < nf; ++bf) {
            if (h->re.fnode[(size_t)0 * nf + bf] == b) v -= cfold * h->re.L[((size_t)0 * nd + a) * nf + bf];
            if (h->re.fnode[(size_t)(r + 1) * nf + bf] == b) v += cfold * h->re.L[((size_t)(r + 1) * nd + a) * nf + bf];
          }
          Dt[((size_t)r * nd + a) * nd + b] = v;
        }
    for (size_t i = 0; i < Lt.size(); ++i) Lt[i] = h->re.L[i];
  }
  HIPCHECK(h, hipMalloc((void**)&h->Dt, Dt.size() * sizeof(double)));
  HIPCHECK(h, hipMalloc((void**)&h->Lt, Lt.size() * sizeof(double)));
  HIPCHECK(h, hipMalloc((void**)&h->md_dev, sizeof(MeshDev)));
  HIPCHECK(h, hipMemcpy(h->Dt, Dt.data(), Dt.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->Lt, Lt.data(), Lt.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->md_dev, &h->md, sizeof(MeshDev), hipMemcpyHostToDevice));

  for (int f = 0; f < 4; ++f) {
    size_t comps = field_is_stress(f) ? (size_t)d * d : (size_t)d;
    h->field_len[f] = (size_t)h->ncells * nd * comps;
    h->field_alloc[f] = (size_t)h->md.ncube_pad * h->ncls * nd * comps;
    const size_t es = h->f32 ? sizeof(float) : sizeof(double);
    if (hipMalloc((void**)&h->field[f], h->field_alloc[f] * es) != hipSuccess)
      return fail(h, SG_ERR_NOMEM, "hipMalloc of a field buffer failed");
    HIPCHECK(h, hipMemset(h->field[f], 0, h->field_alloc[f] * es));
  }
  if (h->use_mfma && h->f32) {
    std::vector<float> fF = mfma32_frags_F(h->re), fG = mfma32_frags_G(h->re), fL = mfma32_frags_L(h->re);
    HIPCHECK(h, hipMalloc((void**)&h->fragF, fF.size() * sizeof(float)));
    HIPCHECK(h, hipMalloc((void**)&h->fragG, fG.size() * sizeof(float)));
    HIPCHECK(h, hipMalloc((void**)&h->fragL, fL.size() * sizeof(float)));
    HIPCHECK(h, hipMemcpy(h->fragF, fF.data(), fF.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragG, fG.data(), fG.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragL, fL.data(), fL.size() * sizeof(float), hipMemcpyHostToDevice));
  } else if (h->use_mfma || h->use_tile) {
    std::vector<double> fF, fG, fL;
    if (h->use_tile) {
      fF = tile2d_frags_V(h->re, -1.0);
      fG = tile2d_frags_V(h->re, 1.0);
      fL = tile2d_frags_L(h->re);
    } else {
      fF = mfma_frags_F(h->re);
      fG = mfma_frags_G(h->re);
      fL = mfma_frags_L(h->re);
    }
    HIPCHECK(h, hipMalloc((void**)&h->fragF, fF.size() * sizeof(double)));
    HIPCHECK(h, hipMalloc((void**)&h->fragG, fG.size() * sizeof(double)));
    HIPCHECK(h, hipMalloc((void**)&h->fragL, fL.size() * sizeof(double)));
    HIPCHECK(h, hipMemcpy(h->fragF, fF.data(), fF.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragG, fG.data(), fG.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragL, fL.data(), fL.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (h->use_mfma || h->use_lane || h->use_tile) {
    // symmetric-stress mode: fields start at zero, g only produces symmetric tensors;
    // left for good as soon as the user uploads a non-symmetric stress or source (SEIGEN_HIP_SYM=0: never entered)
    const char* sym_env = std::getenv("SEIGEN_HIP_SYM");
    h->sym = !(sym_env && std::strcmp(sym_env, "0") == 0);
    HIPCHECK(h, hipMalloc((void**)&h->sym_flag, sizeof(int)));
    HIPCHECK(h, hipMemset(h->sym_flag, 0, sizeof(int)));
  }
  if (std::getenv("SEIGEN_HIP_STAMPS")) {
    HIPCHECK(h, hipMalloc((void**)&h->dbg, 32 * sizeof(unsigned long long)));
    HIPCHECK(h, hipMemset(h->dbg, 0, 32 * sizeof(unsigned long long)));
  }
  {
    // Persistent grid of the MFMA stage kernels: two blocks per CU fill every CU (registers and
    // LDS allow exactly two).  A block with halo neighbours leaves 1/16 of those slots empty, so
    // that RCCL's send/receive kernels can start WHILE an interior launch runs: behind a full
    // grid they only start when it drains (tools/overlap_probe.py), and a stage kernel that found
    // some of its own slots taken would run the late blocks' static shares one after the other.
    hipDeviceProp_t prop;
    HIPCHECK(h, hipGetDeviceProperties(&prop, cfg->device));
    const int slots = (h->use_mfma ? mfma_blocks_per_cu(cfg->degree, h->f32) : 2) * prop.multiProcessorCount;
    h->grid_full = slots / 8 * 8;
    h->grid_blocks = (cfg->nbr_mask != 0 ? slots - slots / 16 : slots) / 8 * 8;
    if (const char* gb = std::getenv("SEIGEN_HIP_GRID_BLOCKS")) h->grid_blocks = std::max(8, std::atoi(gb) / 8 * 8);
    if (cfg->nbr_mask == 0) h->grid_full = h->grid_blocks;
    // 251 blocks per XCD label: with an odd (prime) stride of 4 * 251 items a wave's items do not keep falling on
    // the same column of the mesh, e.g. on the sponge strips at both ends of every row (config 2: 0.240 -> 0.232 ms)
    h->tile_grid = 2008;
    if (const char* tg = std::getenv("SEIGEN_HIP_TILE_GRID")) h->tile_grid = std::max(8, std::atoi(tg) / 8 * 8);
  }
  {
    const char* ge = std::getenv("SEIGEN_HIP_GRAPH");  // 0/1 overrides (measurements)
    const int64_t dofs = h->ncells * (int64_t)h->re.nd * (cfg->dim + cfg->dim * cfg->dim);
    h->graph_ok = ge ? (std::strcmp(ge, "0") != 0) : (!h->use_mfma && dofs <= (int64_t)1 << 23);
  }
  {
    const char* ov = std::getenv("SEIGEN_HIP_OVERLAP");
    h->overlap = cfg->nbr_mask != 0 && !(ov && std::strcmp(ov, "0") == 0);
  }
  int prio_lo = 0, prio_hi = 0;
  HIPCHECK(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  if (cfg->stream) {
    h->stream = (hipStream_t)cfg->stream;
  } else {
    HIPCHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, h->overlap ? prio_hi : 0));
    h->own_stream = true;
  }
  if (h->overlap) {
    HIPCHECK(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_second, hipEventDisableTiming));
  }
  HIPCHECK(h, hipEventCreate(&h->ev0));
  HIPCHECK(h, hipEventCreate(&h->ev1));
  HIPCHECK(h, hipDeviceSynchronize());
  return SG_OK;
}

int sg_create(const sg_config* cfg, sg_handle** out) {
  if (!cfg || !out) {
    g_create_err = "null argument";
    return SG_ERR_ARG;
  }
  *out = nullptr;
  sg_handle* h = new sg_handle();
  int rc = create_impl(cfg, h);
  if (rc != SG_OK) {
    g_create_err = h->err;
    sg_destroy(h);
    return rc;
  }
  *out = h;
  return SG_OK;
}

int sg_get_stream(const sg_handle* h, void** stream) {
  if (!h || !stream) return SG_ERR_ARG;
  *stream = (void*)h->stream;
  return SG_OK;
}

int sg_get_second_stream(const sg_handle* h, void** stream) {
  if (!h || !stream) return SG_ERR_ARG;
  *stream = (void*)h->stream2;
  return SG_OK;
}

int sg_get_info(const sg_handle* h, sg_info_t* out) {
  if (!h || !out) return SG_ERR_ARG;
  const int d = h->cfg.dim;
  out->dim = d;
  out->degree = h->cfg.degree;
  out->nd = h->re.nd;
  out->nf = h->re.nf;
  out->nfaces = h->re.nfaces;
  out->nclasses = h->ncls;
  out->ncells = h->ncells;
  out->u_dofs = (int64_t)d * h->re.nd * h->ncells;
  out->s_dofs = (int64_t)d * d * h->re.nd * h->ncells;
  for (int s = 0; s < 6; ++s) {
    int axis = s >> 1;
    if (axis >= d) {
      out->halo_faces[s] = 0;
      continue;
    }
    int64_t n2 = 1;
    for (int a = 0; a < 3; ++a)
      if (a != axis) n2 *= h->cfg.n[a];
    out->halo_faces[s] = (int32_t)(n2 * h->md.halo_per_cube);
  }
  return SG_OK;
}

int sg_sync(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  return SG_OK;
}

int sg_block_node_coords(const sg_config* cfg, int degree, double* out, size_t nbytes) {
  if (!cfg || !out || degree < 1 || degree > 8 || cfg->dim < 1 || cfg->dim > 3) return SG_ERR_ARG;
  const int d = cfg->dim;
  std::vector<int> lat;
  lattice_points(d, degree, lat);
  const int nq = num_nodes(d, degree);
  int off[MAX_CLS][4][3], ncls;
  class_vertices(d, cfg->diagonal, ncls, off);
  int n[3] = {1, 1, 1};
  for (int a = 0; a < d; ++a) n[a] = cfg->n[a];
  if (nbytes != (size_t)n[0] * n[1] * n[2] * ncls * nq * d * sizeof(double)) return SG_ERR_ARG;
  size_t o = 0;
  for (int ck = 0; ck < n[2]; ++ck)
    for (int cj = 0; cj < n[1]; ++cj)
      for (int ci = 0; ci < n[0]; ++ci) {
        int c[3] = {ci, cj, ck};
        for (int k = 0; k < ncls; ++k) {
          double X[4][3];
          for (int v = 0; v <= d; ++v)
            for (int i = 0; i < d; ++i) X[v][i] = cfg->origin[i] + (double)(c[i] + off[k][v][i]) * cfg->h[i];
          for (int a = 0; a < nq; ++a)
            for (int i = 0; i < d; ++i) {
              double x = X[0][i];
              for (int m = 0; m < d; ++m) x += (X[m + 1][i] - X[0][i]) * ((double)lat[a * d + m] / (double)degree);
              out[o++] = x;
            }
        }
      }
  return SG_OK;
}

int sg_node_coords(const sg_handle* h, int degree, double* out, size_t nbytes) {
  if (!h) return SG_ERR_ARG;
  return sg_block_node_coords(&h->cfg, degree, out, nbytes);
}

int sg_set_params(sg_handle* h, double density, double dt, const double* lambda, const double* mu, int per_cell) {
  if (h) h->epoch += 1;
  if (!h || !lambda || !mu) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  h->rho = density;
  h->rho_physical = 0;
  if (h->rho2_d) {
    HIPCHECK(h, sync_all(h));
    (void)hipFree(h->rho2_d);
    h->rho2_d = nullptr;
  }
  h->dt = dt;
  h->per_cell = per_cell ? 1 : 0;
  if (per_cell) {
    size_t nb = (size_t)h->ncells * sizeof(double);
    if (!h->lam_d) HIPCHECK(h, hipMalloc((void**)&h->lam_d, nb));
    if (!h->mu_d) HIPCHECK(h, hipMalloc((void**)&h->mu_d, nb));
    HIPCHECK(h, sync_all(h));
    HIPCHECK(h, hipMemcpy(h->lam_d, lambda, nb, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->mu_d, mu, nb, hipMemcpyHostToDevice));
    h->lam0 = lambda[0];
    h->mu0 = mu[0];
  } else {
    h->lam0 = lambda[0];
    h->mu0 = mu[0];
  }
  h->params_set = true;
  return SG_OK;
}

int sg_set_density(sg_handle* h, const double* rho, int per_cell, int physical) {
  if (h) h->epoch += 1;
  if (!h || !rho) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  if (h->rho2_d) {
    (void)hipFree(h->rho2_d);
    h->rho2_d = nullptr;
  }
  h->rho_physical = physical ? 1 : 0;
  if (!per_cell) {
    if (physical && rho[0] == 0.0) return fail(h, SG_ERR_ARG, "sg_set_density: zero density");
    h->rho = rho[0];
    return SG_OK;
  }
  std::vector<double> r2((size_t)h->ncells * 2);
  for (int64_t e = 0; e < h->ncells; ++e) {
    if (physical && rho[e] == 0.0) return fail(h, SG_ERR_ARG, "sg_set_density: zero density");
    r2[2 * (size_t)e] = physical ? 1.0 : rho[e];
    r2[2 * (size_t)e + 1] = physical ? 1.0 / rho[e] : 1.0;
  }
  h->rho = rho[0];
  HIPCHECK(h, hipMalloc((void**)&h->rho2_d, r2.size() * sizeof(double)));
  HIPCHECK(h, hipMemcpy(h->rho2_d, r2.data(), r2.size() * sizeof(double), hipMemcpyHostToDevice));
  return SG_OK;
}

// Leave symmetric-stress mode: make the (i > j) lines of both stress buffers valid again.
static int leave_sym_mode(sg_handle* h) {
  h->epoch += 1;
  if (!h->sym) return SG_OK;
  for (int f : {SG_FIELD_S, SG_FIELD_SH})
    if (launch_mirror(h->md, h->field[f], h->f32, h->stream) != 0) return fail(h, SG_ERR_DEVICE, "mirror kernel launch failed");
  HIPCHECK(h, sync_all(h));
  h->sym = false;
  return SG_OK;
}

// host-to-host copy on several threads (one thread moves about 10 GB/s, the PCIe link 50+)
static void parallel_memcpy(void* dst, const void* src, size_t nbytes) {
  unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nbytes < ((size_t)4 << 20)) nt = 1;
  if (nt == 1) {
    std::memcpy(dst, src, nbytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (nbytes / nt + 4095) & ~(size_t)4095;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= nbytes) break;
    const size_t n = std::min(per, nbytes - o);
    th.emplace_back([=]() { std::memcpy((char*)dst + o, (const char*)src + o, n); });
  }
  for (auto& t : th) t.join();
}

static constexpr size_t XFER_CHUNK_BYTES = (size_t)64 << 20;

// Large transfers (sg_set_field / sg_get_field of hundreds of MB): chunks of 64 MB go through pinned
// host slots.  Download: layout kernel -> device slot -> async DMA -> pinned slot, while the host
// copies the previous pinned slot into the caller's (pageable) array on several threads.  Upload: the
// mirror image.  A plain hipMemcpy to pageable memory runs at 11 GB/s (one staging thread inside the
// runtime); this pipeline is bound by the link.
static int transfer_pipelined(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, bool to_device) {
  const size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  const int comps = (int)(per_cell / h->re.nd);
  const size_t chunk_cells = std::max<size_t>(1, XFER_CHUNK_BYTES / (per_cell * sizeof(double)));
  const size_t slot_len = chunk_cells * per_cell;
  for (int i = 0; i < 2; ++i) {
    if (!h->pin[i]) HIPCHECK(h, hipHostMalloc((void**)&h->pin[i], slot_len * sizeof(double), hipHostMallocDefault));
    if (!h->dstage[i] && h->md.gw != 1) HIPCHECK(h, hipMalloc((void**)&h->dstage[i], slot_len * sizeof(double)));
    if (!h->xfer_ev[i]) HIPCHECK(h, hipEventCreateWithFlags(&h->xfer_ev[i], hipEventDisableTiming));
  }
  const int64_t nchunks = (ncells + (int64_t)chunk_cells - 1) / (int64_t)chunk_cells;
  auto range = [&](int64_t c, int64_t& c0, int64_t& n) {
    c0 = c * (int64_t)chunk_cells;
    n = std::min<int64_t>((int64_t)chunk_cells, ncells - c0);
  };
  int* flag = (to_device && h->sym && field_is_stress(field)) ? h->sym_flag : nullptr;
  const int symdl = (!to_device && h->sym && field_is_stress(field)) ? 1 : 0;
  for (int64_t c = 0; c <= nchunks; ++c) {
    const int sl = (int)(c & 1);
    int64_t c0, n;
    if (to_device) {
      if (c < nchunks) {
        range(c, c0, n);
        const size_t nb = (size_t)n * per_cell * sizeof(double);
        HIPCHECK(h, hipEventSynchronize(h->xfer_ev[sl]));  // the slot's previous DMA has left the pinned buffer
        parallel_memcpy(h->pin[sl], host + (size_t)c0 * per_cell, nb);
        if (h->md.gw == 1) {
          HIPCHECK(h, hipMemcpyAsync(h->field[field] + (size_t)(cell0 + c0) * per_cell, h->pin[sl], nb, hipMemcpyHostToDevice, h->stream));
        } else {
          HIPCHECK(h, hipMemcpyAsync(h->dstage[sl], h->pin[sl], nb, hipMemcpyHostToDevice, h->stream));
          if (launch_layout(h->md, comps, 0, h->field[field], h->dstage[sl], cell0 + c0, n, 0, flag, h->f32, h->stream) != 0)
            return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
        }
        HIPCHECK(h, hipEventRecord(h->xfer_ev[sl], h->stream));
      }
    } else {
      if (c < nchunks) {
        range(c, c0, n);
        const size_t nb = (size_t)n * per_cell * sizeof(double);
        if (h->md.gw == 1) {
          HIPCHECK(h, hipMemcpyAsync(h->pin[sl], h->field[field] + (size_t)(cell0 + c0) * per_cell, nb, hipMemcpyDeviceToHost, h->stream));
        } else {
          if (launch_layout(h->md, comps, 1, h->field[field], h->dstage[sl], cell0 + c0, n, symdl, nullptr, h->f32, h->stream) != 0)
            return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
          HIPCHECK(h, hipMemcpyAsync(h->pin[sl], h->dstage[sl], nb, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHECK(h, hipEventRecord(h->xfer_ev[sl], h->stream));
      }
      if (c > 0) {  // the previous chunk has arrived (or is arriving) in the other slot: hand it to the caller
        range(c - 1, c0, n);
        HIPCHECK(h, hipEventSynchronize(h->xfer_ev[sl ^ 1]));
        parallel_memcpy(host + (size_t)c0 * per_cell, h->pin[sl ^ 1], (size_t)n * per_cell * sizeof(double));
      }
    }
  }
  HIPCHECK(h, sync_all(h));
  return SG_OK;
}

// Copy `ncells` cells from `cell0` between a host array in the reference layout and the device
// field.  gw == 1: the layouts coincide; otherwise go through a staging buffer + layout kernel.
static int transfer(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, bool to_device) {
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  const size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  // downloads only: uploads from pageable memory already run at 47 GB/s inside the runtime (measured,
  // tools/transfer_rate.py: 35 GB/s through this pipeline)
  if (!to_device && (size_t)ncells * per_cell * sizeof(double) >= ((size_t)16 << 20) && !std::getenv("SEIGEN_HIP_PLAIN_COPY")) {
    int rc = transfer_pipelined(h, field, cell0, ncells, host, to_device);
    if (rc != SG_OK) return rc;
    if (to_device && h->sym && field_is_stress(field)) {
      int flag = 0;
      HIPCHECK(h, hipMemcpy(&flag, h->sym_flag, sizeof(int), hipMemcpyDeviceToHost));
      if (flag) {
        rc = leave_sym_mode(h);
        if (rc != SG_OK) return rc;
        return transfer_pipelined(h, field, cell0, ncells, host, true);
      }
    }
    return SG_OK;
  }
  if (h->md.gw == 1) {
    double* dev = h->field[field] + (size_t)cell0 * per_cell;
    size_t nb = (size_t)ncells * per_cell * sizeof(double);
    if (to_device)
      HIPCHECK(h, hipMemcpy(dev, host, nb, hipMemcpyHostToDevice));
    else
      HIPCHECK(h, hipMemcpy(host, dev, nb, hipMemcpyDeviceToHost));
    return SG_OK;
  }
  const size_t cap_cells = std::max<size_t>(1, ((size_t)32 << 20) / per_cell);  // 256 MB staging
  if (!h->staging) {
    h->staging_len = std::min(cap_cells, (size_t)h->ncells) * per_cell;
    HIPCHECK(h, hipMalloc((void**)&h->staging, h->staging_len * sizeof(double)));
  }
  const size_t chunk = h->staging_len / per_cell;
  const int comps = (int)(per_cell / h->re.nd);
  for (int64_t done = 0; done < ncells; done += (int64_t)chunk) {
    int64_t n = std::min<int64_t>((int64_t)chunk, ncells - done);
    size_t nb = (size_t)n * per_cell * sizeof(double);
    if (to_device) {
      HIPCHECK(h, hipMemcpy(h->staging, host + (size_t)done * per_cell, nb, hipMemcpyHostToDevice));
      int* flag = (h->sym && field_is_stress(field)) ? h->sym_flag : nullptr;
      if (launch_layout(h->md, comps, 0, h->field[field], h->staging, cell0 + done, n, 0, flag, h->f32, h->stream) != 0)
        return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
      HIPCHECK(h, sync_all(h));
    } else {
      const int symdl = (h->sym && field_is_stress(field)) ? 1 : 0;
      if (launch_layout(h->md, comps, 1, h->field[field], h->staging, cell0 + done, n, symdl, nullptr, h->f32, h->stream) != 0)
        return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
      HIPCHECK(h, sync_all(h));
      HIPCHECK(h, hipMemcpy(host + (size_t)done * per_cell, h->staging, nb, hipMemcpyDeviceToHost));
    }
  }
  if (to_device && h->sym && field_is_stress(field)) {
    int flag = 0;
    HIPCHECK(h, hipMemcpy(&flag, h->sym_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
      // a non-symmetric stress arrived: make every (i > j) line of both stress buffers valid (they
      // are stale wherever kernels ran in symmetric mode), then repeat this upload in full mode
      int rc = leave_sym_mode(h);
      if (rc != SG_OK) return rc;
      return transfer(h, field, cell0, ncells, host, true);
    }
  }
  return SG_OK;
}

int sg_set_field(sg_handle* h, int field, const double* host, size_t nbytes) {
  if (!h || !host || field < 0 || field > 3) return SG_ERR_ARG;
  if (nbytes != h->field_len[field] * sizeof(double)) return fail(h, SG_ERR_ARG, "sg_set_field: size mismatch");
  return transfer(h, field, 0, h->ncells, const_cast<double*>(host), true);
}

int sg_get_field(sg_handle* h, int field, double* host, size_t nbytes) {
  if (!h || !host || field < 0 || field > 3) return SG_ERR_ARG;
  if (nbytes != h->field_len[field] * sizeof(double)) return fail(h, SG_ERR_ARG, "sg_get_field: size mismatch");
  return transfer(h, field, 0, h->ncells, host, false);
}

static int field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, size_t nbytes, const char* who) {
  if (!h || field < 0 || field > 3) return SG_ERR_ARG;
  if (cell0 < 0 || ncells < 0 || cell0 + ncells > h->ncells) return fail(h, SG_ERR_ARG, std::string(who) + ": cell range out of bounds");
  size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  if (nbytes != (size_t)ncells * per_cell * sizeof(double)) return fail(h, SG_ERR_ARG, std::string(who) + ": size mismatch");
  return SG_OK;
}

int sg_set_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, const double* host, size_t nbytes) {
  int rc = field_range(h, field, cell0, ncells, nbytes, "sg_set_field_range");
  if (rc != SG_OK || !host) return rc != SG_OK ? rc : SG_ERR_ARG;
  return transfer(h, field, cell0, ncells, const_cast<double*>(host), true);
}

int sg_get_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, size_t nbytes) {
  int rc = field_range(h, field, cell0, ncells, nbytes, "sg_get_field_range");
  if (rc != SG_OK || !host) return rc != SG_OK ? rc : SG_ERR_ARG;
  return transfer(h, field, cell0, ncells, host, false);
}

int sg_set_absorption(sg_handle* h, const double* sigma_nodes, int sigma_degree) {
  if (h) h->epoch += 1;
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  if (h->sponge_slot) {
    (void)hipFree(h->sponge_slot);
    h->sponge_slot = nullptr;
  }
  if (h->sponge_B) {
    (void)hipFree(h->sponge_B);
    h->sponge_B = nullptr;
  }
  if (!sigma_nodes) return SG_OK;
  if (sigma_degree < 1 || sigma_degree > 6) return fail(h, SG_ERR_ARG, "sigma_degree must be 1..6");
  const int d = h->cfg.dim, nd = h->re.nd;
  const int nq = num_nodes(d, sigma_degree);
  // B_e[a][b] = sum_c A[a][c][b] sigma_{e,c}  for cells with a non-zero sigma
  std::vector<double> A = sponge_tensor(d, h->cfg.degree, sigma_degree);
  std::vector<int32_t> slot((size_t)h->ncells, -1);
  std::vector<double> B;
  int32_t nslots = 0;
  for (int64_t e = 0; e < h->ncells; ++e) {
    const double* sg_ = sigma_nodes + (size_t)e * nq;
    bool nz = false;
    for (int c = 0; c < nq; ++c) nz = nz || (sg_[c] != 0.0);
    if (!nz) continue;
    slot[e] = nslots++;
    size_t base = B.size();
    B.resize(base + (size_t)nd * nd, 0.0);
    for (int a = 0; a < nd; ++a)
      for (int c = 0; c < nq; ++c) {
        double s = sg_[c];
        if (s == 0.0) continue;
        const double* Arow = &A[((size_t)a * nq + c) * nd];
        double* Brow = &B[base + (size_t)a * nd];
        for (int b = 0; b < nd; ++b) Brow[b] += Arow[b] * s;
      }
  }
  HIPCHECK(h, hipMalloc((void**)&h->sponge_slot, slot.size() * sizeof(int32_t)));
  HIPCHECK(h, hipMemcpy(h->sponge_slot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (nslots > 0) {
    HIPCHECK(h, hipMalloc((void**)&h->sponge_B, B.size() * sizeof(double)));
    HIPCHECK(h, hipMemcpy(h->sponge_B, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice));
  } else {
    HIPCHECK(h, hipMalloc((void**)&h->sponge_B, sizeof(double)));
  }
  return SG_OK;
}

struct Box {
  int o[3], n[3];
};
static void region_boxes(int d, const int32_t n[3], const int32_t has_nbr[6], int region, std::vector<Box>& out);
static void region_boxes(const sg_handle* h, int region, std::vector<Box>& out) {
  region_boxes(h->cfg.dim, h->cfg.n, h->md.has_nbr, region, out);
}

int sg_set_source(sg_handle* h, int64_t nnz, const int64_t* nodes, int64_t nsteps, const double* values) {
  if (h) h->epoch += 1;
  if (!h || nnz < 0) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  if (h->src_nodes) {
    (void)hipFree(h->src_nodes);
    h->src_nodes = nullptr;
  }
  if (h->src_values) {
    (void)hipFree(h->src_values);
    h->src_values = nullptr;
  }
  if (h->src_slot_d) {
    (void)hipFree(h->src_slot_d);
    h->src_slot_d = nullptr;
  }
  if (h->src_idx_d) {
    (void)hipFree(h->src_idx_d);
    h->src_idx_d = nullptr;
  }
  h->src_fused = false;
  h->src_nnz = 0;
  h->src_nsteps = 0;
  h->src_step = 0;
  h->src_static = false;
  if (nnz == 0 || nsteps == 0) return SG_OK;
  if (!nodes || !values || nsteps < -1) return SG_ERR_ARG;
  const bool is_static = nsteps == -1;
  if (is_static) nsteps = 1;
  const int d = h->cfg.dim;
  int64_t nscalar = h->ncells * h->re.nd;
  for (int64_t k = 0; k < nnz; ++k)
    if (nodes[k] < 0 || nodes[k] >= nscalar) return fail(h, SG_ERR_ARG, "sg_set_source: node index out of range");
  if (h->sym) {
    bool symmetric = true;
    for (int64_t i = 0; i < nsteps * nnz && symmetric; ++i)
      for (int a = 0; a < d; ++a)
        for (int b = a + 1; b < d; ++b) symmetric = symmetric && values[i * d * d + a * d + b] == values[i * d * d + b * d + a];
    if (!symmetric) {
      int rc = leave_sym_mode(h);
      if (rc != SG_OK) return rc;
    }
  }
  size_t vbytes = (size_t)nsteps * nnz * d * d * sizeof(double);
  HIPCHECK(h, hipMalloc((void**)&h->src_nodes, (size_t)nnz * sizeof(int64_t)));
  HIPCHECK(h, hipMalloc((void**)&h->src_values, vbytes));
  // Order the nodes so that those in cells of SG_REGION_FIRST come first: a split stage adds the
  // source to each part right after the launch that wrote it (the traces of FIRST are packed
  // before SECOND has run).  Then: device offset of component 0 of each node in the field layout.
  std::vector<int64_t> order((size_t)nnz), offs((size_t)nnz);
  {
    const int64_t nd = h->re.nd, ncls = h->ncls, gw = h->md.gw, nc = (int64_t)d * d;
    std::vector<Box> first;
    region_boxes(h, SG_REGION_FIRST, first);
    auto in_first = [&](int64_t node) {
      const int64_t cube = node / nd / ncls;
      const int64_t c[3] = {cube % h->cfg.n[0], (cube / h->cfg.n[0]) % h->cfg.n[1], cube / ((int64_t)h->cfg.n[0] * h->cfg.n[1])};
      for (const Box& b : first) {
        bool in = true;
        for (int k = 0; k < 3; ++k) in = in && c[k] >= b.o[k] && c[k] < b.o[k] + b.n[k];
        if (in) return true;
      }
      return false;
    };
    int64_t n1 = 0;
    for (int64_t i = 0; i < nnz; ++i)
      if (in_first(nodes[i])) order[(size_t)n1++] = i;
    h->src_nfirst = n1;
    for (int64_t i = 0; i < nnz; ++i)
      if (!in_first(nodes[i])) order[(size_t)n1++] = i;
    for (int64_t j = 0; j < nnz; ++j) {
      const int64_t node = nodes[order[(size_t)j]];
      int64_t e = node / nd, b = node % nd;
      int64_t cube = e / ncls, cls = e % ncls;
      offs[(size_t)j] = ((((cube / gw) * ncls + cls) * nd + b) * nc) * gw + cube % gw;
    }
  }
  std::vector<double> vals((size_t)nsteps * nnz * d * d);
  for (int64_t k = 0; k < nsteps; ++k)
    for (int64_t j = 0; j < nnz; ++j)
      std::memcpy(&vals[((size_t)k * nnz + j) * d * d], &values[((size_t)k * nnz + order[(size_t)j]) * d * d], sizeof(double) * d * d);
  HIPCHECK(h, hipMemcpy(h->src_nodes, offs.data(), (size_t)nnz * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->src_values, vals.data(), vbytes, hipMemcpyHostToDevice));
  if (h->use_tile && !std::getenv("SEIGEN_HIP_SOURCE_LAUNCH")) {
    // tile kernels: item (16 squares of one class) -> slot, and per slot a dense (node, cell) -> value-row table, so
    // that the G stages add the source themselves (one launch less per G stage).  A node listed twice keeps the
    // separate launch (which adds both entries).
    const int64_t nd = h->re.nd, nitems = h->md.ncube_pad / 16 * 2;
    std::vector<int32_t> slot((size_t)nitems, -1), idx;
    bool dup = false;
    for (int64_t j = 0; j < nnz && !dup; ++j) {
      const int64_t node = nodes[order[(size_t)j]];
      const int64_t e = node / nd, b = node % nd, cube = e / 2, cls = e % 2;
      const int64_t item = (cube / 16) * 2 + cls;
      if (slot[(size_t)item] < 0) {
        slot[(size_t)item] = (int32_t)(idx.size() / (size_t)(nd * 16));
        idx.resize(idx.size() + (size_t)(nd * 16), -1);
      }
      int32_t& cell = idx[((size_t)slot[(size_t)item] * nd + b) * 16 + cube % 16];
      dup = cell >= 0;
      cell = (int32_t)j;
    }
    if (!dup) {
      HIPCHECK(h, hipMalloc((void**)&h->src_slot_d, slot.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMalloc((void**)&h->src_idx_d, idx.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMemcpy(h->src_slot_d, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HIPCHECK(h, hipMemcpy(h->src_idx_d, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      h->src_fused = true;
    }
  }
  h->src_nnz = nnz;
  h->src_nsteps = nsteps;
  h->src_static = is_static;
  return SG_OK;
}

// ---- stage launches --------------------------------------------------------------------

static void region_boxes(int d, const int32_t n[3], const int32_t has_nbr[6], int region, std::vector<Box>& out) {
  out.clear();
  int lo[3] = {0, 0, 0}, hi[3];
  for (int a = 0; a < 3; ++a) hi[a] = n[a];
  if (region == SG_REGION_ALL) {
    out.push_back(Box{{0, 0, 0}, {hi[0], hi[1], hi[2]}});
    return;
  }
  // interior: peel one cube off every side that has a neighbour block
  int ilo[3] = {0, 0, 0}, ihi[3] = {hi[0], hi[1], hi[2]};
  for (int a = 0; a < d; ++a) {
    if (has_nbr[2 * a]) ilo[a] = 1;
    if (has_nbr[2 * a + 1]) ihi[a] = hi[a] - 1;
    if (ihi[a] < ilo[a]) ihi[a] = ilo[a];
  }
  if (region == SG_REGION_INTERIOR) {
    out.push_back(Box{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}});
    return;
  }
  // FIRST / SECOND: the interior cut in two along the slowest axis (whole runs of the layout)
  const int ax = d - 1, mid = ilo[ax] + (ihi[ax] - ilo[ax]) / 2;
  if (region == SG_REGION_SECOND) {
    Box b{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}};
    b.o[ax] = mid;
    b.n[ax] = ihi[ax] - mid;
    out.push_back(b);
    return;
  }
  if (region == SG_REGION_FIRST) {
    Box b{{ilo[0], ilo[1], ilo[2]}, {ihi[0] - ilo[0], ihi[1] - ilo[1], ihi[2] - ilo[2]}};
    b.n[ax] = mid - ilo[ax];
    out.push_back(b);
  }
  // boundary shell = all \ interior, as disjoint slabs: peel axis by axis
  int clo[3] = {lo[0], lo[1], lo[2]}, chi[3] = {hi[0], hi[1], hi[2]};
  for (int a = 0; a < d; ++a) {
    if (ilo[a] > clo[a]) {
      Box b;
      for (int k = 0; k < 3; ++k) {
        b.o[k] = clo[k];
        b.n[k] = chi[k] - clo[k];
      }
      b.n[a] = ilo[a] - clo[a];
      out.push_back(b);
      clo[a] = ilo[a];
    }
    if (ihi[a] < chi[a] && ihi[a] >= clo[a]) {
      Box b;
      for (int k = 0; k < 3; ++k) {
        b.o[k] = clo[k];
        b.n[k] = chi[k] - clo[k];
      }
      b.o[a] = ihi[a];
      b.n[a] = chi[a] - ihi[a];
      out.push_back(b);
      chi[a] = ihi[a];
    }
  }
}

static bool source_active(const sg_handle* h) {
  return h->src_nnz != 0 && (h->src_static || h->src_step < h->src_nsteps);
}

static int run_op(sg_handle* h, int kind, int in_f, int out_f, int aux_f, int mode, double c_self, double c_aux,
                  double c_new, int region, int uabs_f = SG_FIELD_U, bool with_source = false) {
  StageArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = h->field[in_f];
  a.out = h->field[out_f];
  a.aux = aux_f >= 0 ? h->field[aux_f] : nullptr;
  a.uabs = h->field[uabs_f];
  for (int s = 0; s < 6; ++s) {
    a.ghost[s] = h->ghost[in_f][s];
    // required for interior launches too: masked boundary lanes still form (and load through) the pointer
    if (h->md.has_nbr[s] && !a.ghost[s])
      return fail(h, SG_ERR_STATE, "stage needs a halo buffer that was not attached (sg_halo_attach)");
  }
  a.Dt = h->Dt;
  a.Lt = h->Lt;
  a.md = h->md_dev;
  a.fragV = (kind == 0) ? h->fragF : h->fragG;
  a.fragL = h->fragL;
  a.sym = h->sym ? 1 : 0;
  a.f32 = h->f32;
  a.dbg = h->dbg ? h->dbg + 8 * (kind * 2 + (mode ? 1 : 0)) : nullptr;
  a.sponge_slot = (kind == 0) ? h->sponge_slot : nullptr;
  a.sponge_B = h->sponge_B;
  a.lam = h->lam_d;
  a.mu = h->mu_d;
  a.lam0 = h->lam0;
  a.mu0 = h->mu0;
  a.per_cell = h->per_cell;
  a.rho2 = (kind == 0 && mode == 1) ? h->rho2_d : nullptr;
  a.mode = mode;
  a.c_self = c_self;
  a.c_aux = c_aux;
  a.c_new = c_new;
  if (with_source && h->src_fused && source_active(h)) {  // tile path: the G kernel adds this step's source values
    a.src_slot = h->src_slot_d;
    a.src_idx = h->src_idx_d;
    a.src_vals = h->src_values + (size_t)(h->src_static ? 0 : h->src_step) * h->src_nnz * h->cfg.dim * h->cfg.dim;
  }
  std::vector<Box> boxes;
  region_boxes(h, region, boxes);
  if (h->use_mfma || h->use_lane || h->use_tile) {
    // one launch for the whole region: the kernels scan all cell groups and mask lanes by box
    a.nbox = 0;
    for (const Box& b : boxes) {
      if (b.n[0] <= 0 || b.n[1] <= 0 || b.n[2] <= 0) continue;
      if (a.nbox >= SG_MAX_BOXES) return fail(h, SG_ERR_STATE, "region has more boxes than a launch can carry");
      for (int k = 0; k < 3; ++k) {
        a.boxes_o[a.nbox][k] = b.o[k];
        a.boxes_n[a.nbox][k] = b.n[k];
      }
      a.nbox += 1;
    }
    if (a.nbox == 0) return SG_OK;
    a.spread = (region == SG_REGION_BOUNDARY) ? 1 : 0;
    // only launches that run while an exchange is in flight leave block slots to RCCL
    a.grid_blocks = (region == SG_REGION_INTERIOR || region == SG_REGION_SECOND) ? h->grid_blocks : h->grid_full;
    if (h->use_tile) a.grid_blocks = h->tile_grid;
    a.item_list = nullptr;
    a.nlist = 0;
    if (region != SG_REGION_ALL) {
      // Both regions of a split stage are static: list the (cell group, class) items that have
      // an active cube once.  The interior launch then splits ACTIVE items evenly over the XCDs
      // (a shell is whole z-layers of groups, i.e. the first items of XCD 0 and the last of XCD 7:
      // skipping them inside an even split of all items would leave the launch as long as before);
      // the shell launch deals its few items round-robin over all waves.
      int32_t*& list = h->region_items[region];
      int32_t& nlist = h->region_nitems[region];
      if (nlist < 0) {
        const int64_t gw = h->md.gw, ngroups = h->md.ncube_pad / gw;
        std::vector<char> hit((size_t)ngroups, 0);
        for (int bx = 0; bx < a.nbox; ++bx)
          for (int ck = a.boxes_o[bx][2]; ck < a.boxes_o[bx][2] + a.boxes_n[bx][2]; ++ck)
            for (int cj = a.boxes_o[bx][1]; cj < a.boxes_o[bx][1] + a.boxes_n[bx][1]; ++cj)
              for (int ci = a.boxes_o[bx][0]; ci < a.boxes_o[bx][0] + a.boxes_n[bx][0]; ++ci) {
                int64_t cube = ci + (int64_t)h->cfg.n[0] * (cj + (int64_t)h->cfg.n[1] * ck);
                hit[(size_t)(cube / gw)] = 1;
              }
        std::vector<int32_t> items;
        for (int64_t g = 0; g < ngroups; ++g)
          if (hit[(size_t)g])
            for (int k = 0; k < h->ncls; ++k) items.push_back((int32_t)(g * h->ncls + k));
        nlist = (int32_t)items.size();
        if (!items.empty()) {
          HIPCHECK(h, hipMalloc((void**)&list, items.size() * sizeof(int32_t)));
          HIPCHECK(h, hipMemcpy(list, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
      }
      a.item_list = list;
      a.nlist = nlist;
    }
    int rc = h->use_mfma   ? launch_stage_mfma(kind, h->cfg.degree, a, h->stream)
             : h->use_tile ? launch_stage_tile2d(kind, h->cfg.degree, a, h->t2c, (long)(h->md.ncube_pad / 16) * h->ncls, h->stream)
                           : launch_stage_lane(kind, h->cfg.dim, h->cfg.degree, a, (long)(h->md.ncube_pad / 64) * h->ncls, h->stream);
    if (rc != 0) return fail(h, SG_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    return SG_OK;
  }
  for (const Box& b : boxes) {
    bool empty = false;
    for (int k = 0; k < 3; ++k) {
      a.box_o[k] = b.o[k];
      a.box_n[k] = b.n[k];
      empty = empty || (b.n[k] <= 0);
    }
    if (empty) continue;
    int rc = launch_stage(kind, h->cfg.dim, h->cfg.degree, a, h->stream);
    if (rc != 0) return fail(h, SG_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
  }
  return SG_OK;
}

// the source lives on single nodes: added to each part of a split stage right after the launch
// that wrote it (INTERIOR + BOUNDARY: all of it after the second launch)
static int add_source(sg_handle* h, int field, double coef, int region = SG_REGION_ALL) {
  if (h->src_fused) return SG_OK;  // added by the stage kernel (run_op with_source)
  if (h->src_nnz == 0 || (!h->src_static && h->src_step >= h->src_nsteps) || region == SG_REGION_INTERIOR) return SG_OK;
  const int d = h->cfg.dim;
  int64_t off = 0, cnt = h->src_nnz;
  if (region == SG_REGION_FIRST) cnt = h->src_nfirst;
  if (region == SG_REGION_SECOND) {
    off = h->src_nfirst;
    cnt = h->src_nnz - h->src_nfirst;
  }
  if (cnt == 0) return SG_OK;
  const double* vals = h->src_values + ((size_t)(h->src_static ? 0 : h->src_step) * h->src_nnz + off) * d * d;
  int rc = launch_source(h->field[field], d * d, h->md.gw, cnt, h->src_nodes + off, vals, coef, h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "source kernel launch failed");
  return SG_OK;
}

static int run_stage_impl(sg_handle* h, int stage, int region) {
  const double dt = h->dt, c3 = dt * dt * dt / 24.0;
  int rc = SG_OK;
  switch (stage) {
    case SG_STAGE_UH1:
      return run_op(h, 0, SG_FIELD_S, SG_FIELD_UH, -1, 0, 0, 0, 0, region);
    case SG_STAGE_STEMP:
      rc = run_op(h, 1, SG_FIELD_UH, SG_FIELD_SH, -1, 0, 0, 0, 0, region, SG_FIELD_U, true);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_SH, 1.0, region);
      return rc;
    case SG_STAGE_U1:
      // explicit mode keeps only rhs(form_u1): u1 = rho*u0 + dt*uh1 + dt^3/24*uh2 (elastic.py:341-345, :354-356);
      // sg_set_density(physical = 1): u1 = u0 + (dt*uh1 + dt^3/24*uh2)/rho; per-cell density: factors in rho2
      if (h->rho2_d) return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, 1.0, dt, c3, region);
      if (h->rho_physical) return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, 1.0, dt / h->rho, c3 / h->rho, region);
      return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, h->rho, dt, c3, region);
    case SG_STAGE_SH1:
      rc = run_op(h, 1, SG_FIELD_U, SG_FIELD_SH, -1, 0, 0, 0, 0, region, SG_FIELD_U, true);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_SH, 1.0, region);
      return rc;
    case SG_STAGE_UTEMP:
      return run_op(h, 0, SG_FIELD_SH, SG_FIELD_UH, -1, 0, 0, 0, 0, region);
    case SG_STAGE_S1:
      rc = run_op(h, 1, SG_FIELD_UH, SG_FIELD_S, SG_FIELD_SH, 1, 1.0, dt, c3, region, SG_FIELD_U, true);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_S, c3, region);
      return rc;
  }
  return fail(h, SG_ERR_ARG, "unknown stage");
}

static int resolve_timing(sg_handle* h) {
  if (h->ev_stage_ids.empty()) return SG_OK;
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  for (size_t k = 0; k < h->ev_stage_ids.size(); ++k) {
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[2 * k], h->ev_pool[2 * k + 1]));
    if (h->ev_stage_ids[k] == 6)
      h->counters.halo_pack_ms += ms;
    else
      h->counters.kernel_ms[h->ev_stage_ids[k]] += ms;
  }
  h->ev_stage_ids.clear();
  return SG_OK;
}

int sg_run_stage(sg_handle* h, int stage, int region) {
  if (!h) return SG_ERR_ARG;
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called before stepping");
  if (region < 0 || region > 4) return fail(h, SG_ERR_ARG, "unknown region");
  if (stage < 0 || stage > 5) return fail(h, SG_ERR_ARG, "unknown stage");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const bool second = h->overlap && region == SG_REGION_SECOND;
  hipStream_t const main_stream = h->stream;
  if (second) {
    // depends on everything before this stage's FIRST (ev_stage), not on FIRST itself
    HIPCHECK(h, hipStreamWaitEvent(h->stream2, h->ev_stage, 0));
    h->stream = h->stream2;
  } else {
    if (int rc = join_second(h)) return rc;
    if (h->overlap && region == SG_REGION_FIRST) HIPCHECK(h, hipEventRecord(h->ev_stage, h->stream));
  }
  size_t k = h->ev_stage_ids.size();
  int rc = SG_OK;
  if (h->timing) {
    if (k >= 8192) {
      h->stream = main_stream;
      rc = resolve_timing(h);
      if (rc != SG_OK) return rc;
      if (second) h->stream = h->stream2;
      k = 0;
    }
    while (rc == SG_OK && h->ev_pool.size() < 2 * k + 2) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventCreate failed");
      else h->ev_pool.push_back(e);
    }
    if (rc == SG_OK && hipEventRecord(h->ev_pool[2 * k], h->stream) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
  }
  if (rc == SG_OK) rc = run_stage_impl(h, stage, region);
  if (rc == SG_OK && h->timing) {
    if (hipEventRecord(h->ev_pool[2 * k + 1], h->stream) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
    else h->ev_stage_ids.push_back(stage);
  }
  if (second) {
    if (rc == SG_OK && hipEventRecord(h->ev_second, h->stream2) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
    h->second_pending = rc == SG_OK;
    h->stream = main_stream;
  }
  if (rc != SG_OK) return rc;
  h->counters.launches[stage] += 1;
  return SG_OK;
}

int sg_end_step(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  h->src_step += 1;
  h->counters.steps += 1;
  return SG_OK;
}

// one LF4 step = six launches on the handle's stream (elastic.py:291-304)
static int enqueue_step(sg_handle* h) {
  for (int st = 0; st < 6; ++st) {
    int rc = run_stage_impl(h, st, SG_REGION_ALL);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

// capture `steps` steps into an executable graph; on any failure graphs are switched off for the handle
static hipGraphExec_t capture_steps(sg_handle* h, int steps) {
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
  int rc = SG_OK;
  for (int k = 0; k < steps && rc == SG_OK; ++k) rc = enqueue_step(h);
  hipError_t e = hipStreamEndCapture(h->stream, &g);
  if (rc == SG_OK && e == hipSuccess && g) {
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) ge = nullptr;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return ge;
}

int sg_step(sg_handle* h, int64_t nsteps) {
  if (!h || nsteps < 0) return SG_ERR_ARG;
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called before stepping");
  for (int s = 0; s < 6; ++s)
    if (h->md.has_nbr[s]) return fail(h, SG_ERR_STATE, "sg_step on a block with neighbours: drive stages + halo from the host");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  int64_t k = 0;
  // launch-bound blocks: replay captured graphs (no per-stage timing, no per-step source values)
  const bool graphs = h->graph_ok && !h->timing && h->src_nnz == 0 && nsteps >= 2;
  if (graphs && h->graph_epoch != h->epoch) {
    if (h->graph1) (void)hipGraphExecDestroy(h->graph1);
    if (h->graph8) (void)hipGraphExecDestroy(h->graph8);
    h->graph1 = capture_steps(h, 1);
    h->graph8 = h->graph1 ? capture_steps(h, 8) : nullptr;
    h->graph_epoch = h->epoch;
    if (!h->graph1 || !h->graph8) h->graph_ok = false;  // same kernels, launched one by one below
  }
  HIPCHECK(h, hipEventRecord(h->ev0, h->stream));
  if (graphs && h->graph_ok) {
    for (; k + 8 <= nsteps; k += 8) HIPCHECK(h, hipGraphLaunch(h->graph8, h->stream));
    for (; k < nsteps; ++k) HIPCHECK(h, hipGraphLaunch(h->graph1, h->stream));
    for (int st = 0; st < 6; ++st) h->counters.launches[st] += nsteps;
    h->counters.steps += nsteps;
    h->src_step += nsteps;
  }
  for (; k < nsteps; ++k) {
    for (int st = 0; st < 6; ++st) {
      if (h->timing) {
        int rc = sg_run_stage(h, st, SG_REGION_ALL);
        if (rc != SG_OK) return rc;
      } else {
        int rc = run_stage_impl(h, st, SG_REGION_ALL);
        if (rc != SG_OK) return rc;
        h->counters.launches[st] += 1;
      }
    }
    h->src_step += 1;
    h->counters.steps += 1;
  }
  HIPCHECK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHECK(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  return SG_OK;
}

int sg_last_step_ms(sg_handle* h, double* ms) {
  if (!h || !ms) return SG_ERR_ARG;
  *ms = h->last_ms;
  return SG_OK;
}

int sg_apply_F(sg_handle* h, int s_in, int u_abs, int u_out) {
  if (!h) return SG_ERR_ARG;
  if (!field_is_stress(s_in) || field_is_stress(u_out) || field_is_stress(u_abs) || u_abs == u_out)
    return fail(h, SG_ERR_ARG, "sg_apply_F: s_in must be a stress field, u_abs/u_out distinct velocity fields");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return run_op(h, 0, s_in, u_out, -1, 0, 0, 0, 0, SG_REGION_ALL, u_abs);
}

int sg_apply_G(sg_handle* h, int u_in, int s_out, int use_source) {
  if (!h) return SG_ERR_ARG;
  if (field_is_stress(u_in) || !field_is_stress(s_out))
    return fail(h, SG_ERR_ARG, "sg_apply_G: u_in must be a velocity field, s_out a stress field");
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called first");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  int rc = run_op(h, 1, u_in, s_out, -1, 0, 0, 0, 0, SG_REGION_ALL, SG_FIELD_U, use_source != 0);
  if (rc == SG_OK && use_source) rc = add_source(h, s_out, 1.0);
  return rc;
}

// ---- halo ---------------------------------------------------------------------------------

int sg_halo_bytes(const sg_handle* h, int field, int side, size_t* nbytes) {
  if (!h || !nbytes || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  const int d = h->cfg.dim;
  int axis = side >> 1;
  size_t n2 = 1;
  for (int a = 0; a < 3; ++a)
    if (a != axis) n2 *= (size_t)h->cfg.n[a];
  // dim components per facet node for every field: a stress trace travels as T_i,axis (kernels.hip pack_one)
  *nbytes = n2 * h->md.halo_per_cube * h->re.nf * (size_t)d * (h->f32 ? sizeof(float) : sizeof(double));
  (void)field;
  return SG_OK;
}

// pack launches are timed like stage launches (event pairs resolved lazily; stage id 6 = halo pack)
static int pack_begin(sg_handle* h, size_t& k) {
  if (int rc = join_second(h)) return rc;
  k = h->ev_stage_ids.size();
  if (!h->timing) return SG_OK;
  if (k >= 8192) {
    int rc = resolve_timing(h);
    if (rc != SG_OK) return rc;
    k = 0;
  }
  while (h->ev_pool.size() < 2 * k + 2) {
    hipEvent_t e;
    HIPCHECK(h, hipEventCreate(&e));
    h->ev_pool.push_back(e);
  }
  HIPCHECK(h, hipEventRecord(h->ev_pool[2 * k], h->stream));
  return SG_OK;
}

static int pack_end(sg_handle* h, size_t k, size_t nbytes) {
  if (h->timing) {
    HIPCHECK(h, hipEventRecord(h->ev_pool[2 * k + 1], h->stream));
    h->ev_stage_ids.push_back(6);
  }
  h->counters.halo_pack_launches += 1;
  h->counters.halo_bytes_packed += (int64_t)nbytes;
  return SG_OK;
}

int sg_halo_pack(sg_handle* h, int field, int side, void* dev_out) {
  if (!h || !dev_out || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const int d = h->cfg.dim;
  int comps = field_is_stress(field) ? d * d : d;
  void* out = dev_out;
  size_t k = 0, nb = 0;
  int rc = pack_begin(h, k);
  if (rc != SG_OK) return rc;
  rc = launch_pack(h->md_dev, h->md, h->field[field], comps, 1, &side, &out,
                   (h->sym && field_is_stress(field)) ? 1 : 0, h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "pack kernel launch failed");
  (void)sg_halo_bytes(h, field, side, &nb);
  return pack_end(h, k, nb);
}

int sg_halo_pack_sides(sg_handle* h, int field, void* const* dev_out) {
  if (!h || !dev_out || field < 0 || field > 3) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const int d = h->cfg.dim;
  int comps = field_is_stress(field) ? d * d : d;
  int sides[6], n = 0;
  void* outs[6];
  for (int s = 0; s < 2 * d; ++s)
    if (dev_out[s]) {
      sides[n] = s;
      outs[n] = dev_out[s];
      n += 1;
    }
  size_t k = 0, total = 0;
  int rc = pack_begin(h, k);
  if (rc != SG_OK) return rc;
  rc = launch_pack(h->md_dev, h->md, h->field[field], comps, n, sides, outs,
                   (h->sym && field_is_stress(field)) ? 1 : 0, h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "pack kernel launch failed");
  for (int i = 0; i < n; ++i) {
    size_t nb = 0;
    (void)sg_halo_bytes(h, field, sides[i], &nb);
    total += nb;
  }
  return pack_end(h, k, total);
}

int sg_get_sym(const sg_handle* h, int* sym) {
  if (!h || !sym) return SG_ERR_ARG;
  *sym = h->sym ? 1 : 0;
  return SG_OK;
}

int sg_leave_sym(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return leave_sym_mode(h);
}

int sg_halo_attach(sg_handle* h, int field, int side, const void* dev_in) {
  if (!h || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  h->ghost[field][side] = (const double*)dev_in;
  return SG_OK;
}

// ---- instrumentation ----------------------------------------------------------------------------

int sg_enable_timing(sg_handle* h, int on) {
  if (!h) return SG_ERR_ARG;
  if (!on) {
    int rc = resolve_timing(h);
    if (rc != SG_OK) return rc;
  }
  h->timing = on != 0;
  return SG_OK;
}

int sg_get_counters(sg_handle* h, sg_counters_t* out) {
  if (!h || !out) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  int rc = resolve_timing(h);
  if (rc != SG_OK) return rc;
  *out = h->counters;
  return SG_OK;
}

int64_t sg_reference_operator(int dim, int degree, int which, int q, double* out, size_t nbytes) {
  std::vector<double> v;
  try {
    if (which == 3) {
      if (q < 1 || q > 6 || dim < 1 || dim > 3 || degree < 1 || degree > 4) return SG_ERR_ARG;
      v = sponge_tensor(dim, degree, q);
    } else {
      RefElem re = make_refelem(dim, degree);
      if (which == 0) v = re.D;
      else if (which == 1) v = re.L;
      else if (which == 2) v = re.Mhat;
      else if (which == 4) v.assign(re.fnode.begin(), re.fnode.end());
      else return SG_ERR_ARG;
    }
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return SG_ERR_ARG;
  }
  if (out) {
    if (nbytes != v.size() * sizeof(double)) return SG_ERR_ARG;
    std::memcpy(out, v.data(), nbytes);
  }
  return (int64_t)v.size();
}

int sg_region_boxes(const sg_config* cfg, int region, int32_t* boxes, int max_boxes) {
  if (!cfg || !boxes || cfg->dim < 1 || cfg->dim > 3 || region < 0 || region > 4 || max_boxes < 0) return SG_ERR_ARG;
  int32_t n[3] = {1, 1, 1}, has_nbr[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < cfg->dim; ++a) n[a] = cfg->n[a];
  for (int s2 = 0; s2 < 2 * cfg->dim; ++s2) has_nbr[s2] = (cfg->nbr_mask >> s2) & 1;
  std::vector<Box> out;
  region_boxes(cfg->dim, n, has_nbr, region, out);
  int cnt = 0;
  for (const Box& b : out) {
    if (b.n[0] <= 0 || b.n[1] <= 0 || b.n[2] <= 0) continue;
    if (cnt < max_boxes)
      for (int k = 0; k < 3; ++k) {
        boxes[6 * cnt + k] = b.o[k];
        boxes[6 * cnt + 3 + k] = b.n[k];
      }
    cnt += 1;
  }
  return cnt;
}

int sg_tabulate(int dim, int degree, int64_t npts, const double* xi, double* phi) {
  if (dim < 1 || dim > 3 || degree < 1 || degree > 8 || npts < 0 || !xi || !phi) return SG_ERR_ARG;
  tabulate(dim, degree, (int)npts, xi, phi);
  return SG_OK;
}

int sg_mesh_tables(int dim, int degree, int diagonal, const double* h, int32_t* nb, int32_t* nb_node, double* cn,
                   double* jinv) {
  if (!h || !nb || !nb_node || !cn || !jinv) return SG_ERR_ARG;
  try {
    RefElem re = make_refelem(dim, degree);
    MeshDev md;
    std::memset(&md, 0, sizeof(md));
    md.nd = re.nd;
    md.nf = re.nf;
    double hh[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) hh[a] = h[a];
    build_mesh_tables(dim, degree, diagonal, hh, re.fnode.data(), re.lattice.data(), md);
    for (int c = 0; c < md.ncls; ++c) {
      for (int f = 0; f < md.nfaces; ++f) {
        int32_t* o = nb + ((size_t)c * md.nfaces + f) * 5;
        o[0] = md.nb_axis[c][f];
        o[1] = md.nb_dir[c][f];
        o[2] = md.nb_cls[c][f];
        o[3] = md.nb_face[c][f];
        o[4] = md.face_ord[c][f];
        for (int b = 0; b < md.nf; ++b) nb_node[((size_t)c * md.nfaces + f) * md.nf + b] = md.nb_node[c][f][b];
        for (int j = 0; j < 3; ++j) cn[((size_t)c * md.nfaces + f) * 3 + j] = md.cn[c][f][j];
      }
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 3; ++j) jinv[((size_t)c * 3 + r) * 3 + j] = md.Jinv[c][r][j];
    }
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return SG_ERR_ARG;
  }
  return SG_OK;
}

}  // extern "C"
